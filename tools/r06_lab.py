"""Round-6 scratch lab: timing cases for the split-K block kernel's variants (reads R06_CASE)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import splitk_lab as L  # noqa: E402

f16, bf16 = torch.float16, torch.bfloat16
case = os.environ.get("R06_CASE", "xcd")
tag = os.path.basename(os.environ.get("FLUTE_AMD_LIB", "shipped"))
if case == "xcd":
    for rep in range(2):
        for mb in (1, 2, 4):
            L.time_one(256, 4096, 4096, 4, f16, dict(family=6, splitk=1, kw=4, m_block=mb), tag=f"xcd_group_{mb}")
    for mb in (1, 2, 4, 8):
        L.time_one(512, 2048, 4096, 4, f16, dict(family=6, splitk=1, kw=4, m_block=mb), tag=f"xcd_group_{mb}")
    for mb in (1, 2):
        L.time_one(128, 8192, 4096, 4, f16, dict(family=6, splitk=1, kw=4, m_block=mb), tag=f"xcd_group_{mb}")
elif case == "abl":
    L.time_one(256, 4096, 4096, 4, f16, dict(family=6, splitk=1, kw=4), tag=tag)
    L.time_one(256, 4096, 4096, 4, f16, dict(family=6, splitk=2, kw=2, m_tiles=4), tag=tag)
elif case == "m256":
    L.time_one(256, 4096, 4096, 4, f16, dict(family=6, splitk=1, kw=4), tag=tag)
    L.time_one(256, 4096, 4096, 4, bf16, dict(family=6, splitk=1, kw=4), tag=tag)
    L.time_one(128, 8192, 4096, 4, f16, dict(family=6, splitk=1, kw=4), tag=tag)
    L.time_one(256, 11008, 4096, 4, f16, dict(family=6, splitk=1, kw=2, m_tiles=8), tag=tag)
elif case == "auto":      # the automatic plan against the forced alternatives on shapes the round-6 model extrapolates to
    for (M, N, K) in ((48, 3584, 14336), (64, 8192, 8192), (33, 8192, 8192), (96, 14336, 4096), (80, 8192, 8192), (128, 4096, 4096), (192, 4096, 4096),
                      (256, 4096, 4096), (512, 2048, 4096), (256, 2048, 8192), (128, 11008, 4096), (64, 4096, 4096), (384, 4096, 4096), (512, 4096, 4096)):
        L.time_one(M, N, K, 4, f16, None, tag="tuned table")
        L.time_one(M, N, K, 4, f16, dict(), tag="automatic (id 16)")
        L.time_one(M, N, K, 4, f16, dict(family=2), tag="per-wave")
        L.time_one(M, N, K, 4, f16, dict(family=6, kw=4), tag="kp4 best")
        L.time_one(M, N, K, 4, f16, dict(family=6, kw=2), tag="kp2 best")
