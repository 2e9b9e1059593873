import torch, time
dev=torch.device('cuda:0')
for (B,H,S,D,c) in [(4,32,4096,64,False),(4,32,4096,128,False),(4,32,16384,128,True)]:
    q=torch.randn(B,H,S,D,device=dev,dtype=torch.float16); k=torch.randn_like(q); v=torch.randn_like(q)
    for backend in ['flash','efficient','math']:
        try:
            from torch.nn.attention import sdpa_kernel, SDPBackend
            bk={'flash':SDPBackend.FLASH_ATTENTION,'efficient':SDPBackend.EFFICIENT_ATTENTION,'math':SDPBackend.MATH}[backend]
            if backend=='math' and S>4096: continue
            with sdpa_kernel(bk):
                for _ in range(3): o=torch.nn.functional.scaled_dot_product_attention(q,k,v,is_causal=c)
                torch.cuda.synchronize(); t0=time.perf_counter()
                for _ in range(10): o=torch.nn.functional.scaled_dot_product_attention(q,k,v,is_causal=c)
                torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
            fl=4*B*H*S*S*D/(2 if c else 1)
            print(B,H,S,D,c,backend, f"{dt*1e3:.3f} ms {fl/dt/1e12:.1f} TFLOP/s", flush=True)
        except Exception as e:
            print(backend,'failed',str(e)[:100])
