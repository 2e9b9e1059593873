"""Device layout of lbfa_quant_v_fp8's output (include/lowbit_fa.h: "a device detail, not API"), restated for the tests that
look inside it: [B, H, ceil(S/64), D, 64] e4m3 bytes - per 64-key tile, channel-major; key 32 kb2 + 8 g + 4 hh + e of the tile at
byte 32 hh + 16 kb2 + 4 g + e of its channel's row (the k order of the block-scaled 32x32x64 MFMA's operand, attn_fwd.hip,
vfp8_pos_of_key in quant_kernels.hip), 16-byte chunk c of channel d stored at chunk c ^ ((d >> 2) & 3)."""
import numpy as np


def pos_of_key(key):
    kb2, w = key >> 5, key & 31
    return 32 * ((w >> 2) & 1) + 16 * kb2 + 4 * (w >> 3) + (w & 3)


def chunk_swizzle(d):
    return (d >> 2) & 3


def decode_v_fp8(raw, B, H, S, D):
    """raw: flat uint8 buffer of lbfa_quant_v_fp8 -> codes [B, H, ntile * 64, D] in key order"""
    ntile = (S + 63) // 64
    tiles = np.asarray(raw[: B * H * ntile * D * 64]).reshape(B, H, ntile, D, 64)
    got = np.zeros((B, H, ntile * 64, D), np.uint8)
    for key in range(64):
        pos = pos_of_key(key)
        for d in range(D):
            got[:, :, key::64, d] = tiles[:, :, :, d, (((pos >> 4) ^ chunk_swizzle(d)) << 4) | (pos & 15)]
    return got
