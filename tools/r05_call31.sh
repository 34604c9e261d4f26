#!/bin/bash
# round 5, GPU call 31: M = 256 on 4096^2 - forced variants of the split-K block kernel (loader waves off, bf16, 128-row tiles) next to the automatic plan
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/time_cases.jsonl
timeout 400 python tools/time_cases.py "4,256,4096,4096,f16;4,256,4096,4096,f16,family=6,m_tiles=4,splitk=2;4,256,4096,4096,f16,family=6,m_tiles=4,splitk=2,waves=8;4,256,4096,4096,f16,family=6,m_tiles=8,splitk=4;4,256,4096,4096,f16,family=6,m_tiles=4,splitk=4;4,256,4096,4096,f16,family=2;4,256,4096,4096,bf16;4,256,4096,4096,bf16,family=6,m_tiles=4,splitk=2;4,256,4096,4096,bf16,family=6,m_tiles=4,splitk=2,waves=8;4,200,4096,4096,f16;4,200,4096,4096,f16,family=6,m_tiles=4,splitk=2;4,192,4096,4096,f16;4,192,4096,4096,f16,family=6,m_tiles=4,splitk=2" --steps 400 --tag m256 2>&1 | cut -c1-300
