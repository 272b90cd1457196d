"""Generates the committed golden fixtures (tests/golden/*.npz).

The reference cannot be compiled or imported in this environment and ships no golden vectors
(SURVEY.md §8c), so these vectors come from the oracle the reference's own tests use: torch SDPA in fp32 on
the CPU, applied to seeded fp16/bf16 inputs (stored bit-exactly as uint16).  Each file holds the inputs, the
SDPA-fp32 expected output (`expect`) and the tolerances the tests assert.  Re-run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402


def bits(t):
    return t.contiguous().view(torch.int16).numpy().view(np.uint16)


def rnd(shape, dtype, gen):
    return torch.randn(shape, generator=gen, dtype=torch.float32).to(dtype)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(name, f"{os.path.getsize(path) / 1e6:.2f} MB")


def main():
    # G1 — BASELINE config 1: fp32 B2 S128 H4 D64 non-causal, SDPA on CPU (plumbing/oracle smoke)
    g = torch.Generator().manual_seed(101)
    q, k, v = (torch.randn(2, 128, 4, 64, generator=g) for _ in range(3))
    save("g1_fp32_b2_s128_h4_d64", q=q.numpy(), k=k.numpy(), v=v.numpy(),
         expect=O.sdpa_dense(q, k, v, False).numpy(), atol=np.float32(1e-3), rtol=np.float32(1e-3))

    # G2 — fp16 causal prefill, D128, ragged lengths around the 64-key tile, plus one S=1024 single head
    g = torch.Generator().manual_seed(202)
    arrays = {}
    for i, (S, H, Hk) in enumerate([(64, 2, 2), (65, 2, 2), (127, 4, 2), (320, 2, 1), (1024, 1, 1)]):
        q, k, v = rnd((1, S, H, 128), torch.float16, g), rnd((1, S, Hk, 128), torch.float16, g), rnd((1, S, Hk, 128), torch.float16, g)
        arrays.update({f"q{i}": bits(q), f"k{i}": bits(k), f"v{i}": bits(v), f"expect{i}": O.sdpa_dense(q, k, v, True).numpy()})
    save("g2_fp16_causal_d128", n=np.int32(5), atol=np.float32(1e-3), rtol=np.float32(1e-3), **arrays)

    # G3 — bf16 GQA 3:1 decode (Hq 6 / Hkv 2, the 24:8 ratio of BASELINE config 3), ragged cache lengths
    g = torch.Generator().manual_seed(303)
    B, Sk, H, Hk, D = 2, 600, 6, 2, 128
    q, kc, vc = rnd((B, 1, H, D), torch.bfloat16, g), rnd((B, Sk, Hk, D), torch.bfloat16, g), rnd((B, Sk, Hk, D), torch.bfloat16, g)
    arrays = {"q": bits(q), "k": bits(kc), "v": bits(vc)}
    lens_list = [[1, 63], [64, 65], [600, 321]]
    for i, lens in enumerate(lens_list):
        lt = torch.tensor(lens, dtype=torch.int32)
        arrays[f"lens{i}"] = lt.numpy()
        arrays[f"expect{i}"] = O.sdpa_decode(q, kc, vc, lt).numpy()
    save("g3_bf16_decode_gqa", n=np.int32(len(lens_list)), atol=np.float32(1e-3), rtol=np.float32(1e-3), **arrays)

    # G4 — BASELINE config 4 exactly: varlen fp16 cu_seqlens=[0,128,384,896] H8 D64 causal
    g = torch.Generator().manual_seed(404)
    cu = torch.tensor([0, 128, 384, 896], dtype=torch.int32)
    q, k, v = (rnd((896, 8, 64), torch.float16, g) for _ in range(3))
    save("g4_fp16_varlen_h8_d64", q=bits(q), k=bits(k), v=bits(v), cu=cu.numpy(),
         expect=O.sdpa_varlen(q, k, v, cu, cu, True).numpy(), atol=np.float32(1e-3), rtol=np.float32(1e-3))

    # G5 — bf16 paged decode, pages 16 and 256, permuted block tables, ragged cache lengths
    g = torch.Generator().manual_seed(505)
    arrays = {}
    for i, (page, Sk) in enumerate([(16, 200), (256, 1024)]):
        B, H, Hk, D = 2, 6, 2, 128
        nb = (Sk + page - 1) // page
        table = torch.randperm(B * nb, generator=g).to(torch.int32).view(B, nb)
        kp, vp = rnd((B * nb, page, Hk, D), torch.bfloat16, g), rnd((B * nb, page, Hk, D), torch.bfloat16, g)
        q = rnd((B, 1, H, D), torch.bfloat16, g)
        lens = torch.tensor([Sk, Sk // 2 + 3], dtype=torch.int32)
        arrays.update({f"q{i}": bits(q), f"k{i}": bits(kp), f"v{i}": bits(vp), f"table{i}": table.numpy(),
                       f"lens{i}": lens.numpy(), f"expect{i}": O.sdpa_decode(q, kp, vp, lens, table).numpy()})
    save("g5_bf16_paged_decode", n=np.int32(2), atol=np.float32(1e-3), rtol=np.float32(1e-3), **arrays)


if __name__ == "__main__":
    main()
