#!/bin/bash
# Round 5's GPU calls, one function each (they were 42 one-off scripts tools/r05_call<N>.sh; profiles/README.md names the call that
# produced each log): every function body is the command line that ran on the GPU box through gpurun, from the repo root.
#   tools/r05_gpu_calls.sh call19

call1() {
# round 5, first GPU call: the lean decode kernel in the lab (timing + stamps), the graph-replay probe, the widened GPU suite
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
timeout 150 tools/ubench/oneshot_lab 4096 4096 64 a fast > gpurun_out/r05/lab_fast_run1.jsonl 2>&1
timeout 150 tools/ubench/oneshot_lab_stamps 4096 4096 64 a fast > gpurun_out/r05/lab_fast_stamps_run1.jsonl 2>&1
timeout 100 tools/ubench/oneshot_lab 11008 4096 64 b fast >> gpurun_out/r05/lab_fast_run1.jsonl 2>&1
timeout 100 tools/ubench/oneshot_lab 4096 4096 128 c fast >> gpurun_out/r05/lab_fast_run1.jsonl 2>&1
timeout 60 tools/ubench/oneshot_lab 4096 4096 64 d floors >> gpurun_out/r05/lab_fast_run1.jsonl 2>&1
timeout 200 python tools/graph_probe.py > gpurun_out/r05/graph_probe.json 2> gpurun_out/r05/graph_probe.err
timeout 1100 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r05/pytest_gpu_run1.log 2>&1
tail -5 gpurun_out/r05/pytest_gpu_run1.log
grep -h '"variant"' gpurun_out/r05/lab_fast_run1.jsonl | cut -c1-220
}

call2() {
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
timeout 200 tools/ubench/oneshot_lab 4096 4096 64 a fast > gpurun_out/r05/lab_fast_run2.jsonl 2>&1
timeout 200 tools/ubench/oneshot_lab_stamps 4096 4096 64 a fast > gpurun_out/r05/lab_fast_stamps_run2.jsonl 2>&1
timeout 100 tools/ubench/oneshot_lab 11008 4096 64 b fast >> gpurun_out/r05/lab_fast_run2.jsonl 2>&1
timeout 100 tools/ubench/oneshot_lab 4096 8192 128 c fast >> gpurun_out/r05/lab_fast_run2.jsonl 2>&1
grep -h '"variant"' gpurun_out/r05/lab_fast_run2.jsonl | cut -c1-200
}

call3() {
# round 5, third GPU call: the lean kernel through the library - parity, forced against automatic plans, the bench at 20 / 2000 steps
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
timeout 300 python -m pytest tests/test_qgemm_gpu.py tests/test_abi.py -x -q -m gpu -k "lean or decode_plan or persistent or golden or abi" > gpurun_out/r05/pytest_lean.log 2>&1
tail -3 gpurun_out/r05/pytest_lean.log
rm -f gpurun_out/r05/time_cases_lean.jsonl
C=""
for s in "4096,4096" "11008,4096" "14336,4096" "6144,4096" "1024,4096" "28672,4096" "4096,8192" "8192,8192" "1024,8192" "3584,8192" "4096,2048" "8192,2048"; do
  C="$C;4,1,$s,f16;4,1,$s,f16,one_shot=4"
done
C="$C;4,1,4096,4096,bf16;4,1,4096,4096,bf16,one_shot=4;4,1,4096,8192,bf16;4,1,4096,8192,bf16,one_shot=4"
timeout 400 python tools/time_cases.py "${C:1}" --steps 300 --tag lean --out gpurun_out/r05/time_cases_lean.jsonl > gpurun_out/r05/time_cases_lean.log 2>&1
cat gpurun_out/r05/time_cases_lean.log | cut -c1-260
timeout 200 python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_steps20_run1.json 2> gpurun_out/r05/bench_steps20_run1.err
timeout 200 python bench.py --steps 2000 --warmup 50 --no-extras --no-cpu > gpurun_out/r05/bench_steps2000_run1.json 2> gpurun_out/r05/bench_steps2000_run1.err
python - <<'PY'
import json
for f in ("gpurun_out/r05/bench_steps20_run1.json", "gpurun_out/r05/bench_steps2000_run1.json"):
    try:
        d = json.loads(open(f).readline())
        print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel_us"], d["roofline"]["kernel_us_hip_events"], d["roofline"]["kernel_us_clock"], d.get("m256"), d["eager_us_per_step"])
    except Exception as e:
        print(f, "ERR", e)
PY
}

call4() {
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
timeout 300 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "lean" > gpurun_out/r05/pytest_lean2.log 2>&1
tail -3 gpurun_out/r05/pytest_lean2.log
rm -f gpurun_out/r05/time_cases_lean2.jsonl
C=""
for s in "4096,4096" "5120,4096" "6144,4096" "8192,4096" "11008,4096" "14336,4096" "28672,4096" "2048,4096" "3072,4096"; do
  C="$C;4,1,$s,f16,one_shot=3;4,1,$s,f16,one_shot=4,waves=4;4,1,$s,f16,one_shot=4,waves=8;4,1,$s,f16,one_shot=1"
done
timeout 600 python tools/time_cases.py "${C:1}" --steps 300 --tag lean2 --out gpurun_out/r05/time_cases_lean2.jsonl > gpurun_out/r05/time_cases_lean2.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r05/time_cases_lean2.jsonl"):
    d = json.loads(l)
    print(d["N"], d["K"], d["ovr"], d["us"], d["plan"]["one_shot"], d["plan"]["waves"], d["plan"]["kw"], d["plan"]["grid"])
PY
}

call5() {
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
rm -f gpurun_out/r05/lab_fast_run3.jsonl
timeout 200 tools/ubench/oneshot_lab 4096 4096 64 a fast >> gpurun_out/r05/lab_fast_run3.jsonl 2>&1
timeout 200 tools/ubench/oneshot_lab 4096 4096 64 a2 fast >> gpurun_out/r05/lab_fast_run3.jsonl 2>&1
timeout 100 tools/ubench/oneshot_lab 11008 4096 64 b fast >> gpurun_out/r05/lab_fast_run3.jsonl 2>&1
timeout 100 tools/ubench/oneshot_lab 4096 8192 64 c fast >> gpurun_out/r05/lab_fast_run3.jsonl 2>&1
timeout 100 tools/ubench/oneshot_lab 8192 2048 64 d fast >> gpurun_out/r05/lab_fast_run3.jsonl 2>&1
timeout 200 tools/ubench/oneshot_lab_stamps 4096 4096 64 a fast > gpurun_out/r05/lab_fast_stamps_run3.jsonl 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r05/lab_fast_run3.jsonl"):
    try: d = json.loads(l)
    except Exception: print(l[:150]); continue
    if "variant" in d: print(d.get("tag"), d["variant"], d.get("us"), d.get("rel_err"), d.get("nbad"))
    elif "error" in d: print(d)
PY
}

call6() {
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
timeout 300 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "lean or golden or decode_plan" > gpurun_out/r05/pytest_lean3.log 2>&1
tail -3 gpurun_out/r05/pytest_lean3.log
rm -f gpurun_out/r05/lab_fast_run4.jsonl
timeout 200 tools/ubench/oneshot_lab 4096 4096 64 a fast >> gpurun_out/r05/lab_fast_run4.jsonl 2>&1
timeout 100 tools/ubench/oneshot_lab 11008 4096 64 b fast >> gpurun_out/r05/lab_fast_run4.jsonl 2>&1
timeout 200 tools/ubench/oneshot_lab_stamps 4096 4096 64 a fast > gpurun_out/r05/lab_fast_stamps_run4.jsonl 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r05/lab_fast_run4.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    if "variant" in d and ("fast" in d["variant"] or "pipe" in d["variant"]): print(d.get("tag"), d["variant"], d.get("us"), d.get("rel_err"), d.get("nbad"))
PY
rm -f gpurun_out/r05/time_cases_lean3.jsonl
C=""
for s in "16384,2048" "8192,2048" "4608,2048" "4096,2048"; do
  C="$C;4,1,$s,f16,one_shot=3;4,1,$s,f16,one_shot=4;4,1,$s,f16,one_shot=1"
done
C="$C;4,1,4096,4096,f16;4,1,11008,4096,f16;4,1,8192,4096,f16;4,1,14336,4096,f16"
timeout 300 python tools/time_cases.py "${C:1}" --steps 300 --tag lean3 --out gpurun_out/r05/time_cases_lean3.jsonl > gpurun_out/r05/time_cases_lean3.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r05/time_cases_lean3.jsonl"):
    d = json.loads(l)
    print(d["N"], d["K"], d["ovr"], d["tid"], d["us"], d["plan"]["one_shot"], d["plan"]["waves"], d["plan"]["kw"], d["plan"]["grid"])
PY
cp flute_amd/data/gfx950_tuned.json gpurun_out/r05/tuned_retune_m1.json
timeout 700 python -m flute_amd.tune --shapes "1024,4096;3584,4096;4096,4096;4608,4096;6144,4096;8192,4096;11008,4096;14336,4096;16384,4096;28672,4096;4608,2048;8192,2048;16384,2048" \
    --ms 1 --bits 4 --groups 64,128 --retune --rep 40 --budget-s 600 --out gpurun_out/r05/tuned_retune_m1.json > gpurun_out/r05/retune_m1.log 2>&1
tail -2 gpurun_out/r05/retune_m1.log
}

call7() {
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
cp flute_amd/data/gfx950_tuned.json gpurun_out/r05/tuned_retune_m1.json
timeout 700 python -m flute_amd.tune --shapes "1024,4096;3584,4096;4096,4096;4608,4096;6144,4096;8192,4096;11008,4096;14336,4096;16384,4096;28672,4096;4608,2048;8192,2048;16384,2048" \
    --ms 1 --bits 4 --groups 64,128 --retune --rep 120 --budget-s 600 --out gpurun_out/r05/tuned_retune_m1.json > gpurun_out/r05/retune_m1.log 2>&1
tail -1 gpurun_out/r05/retune_m1.log
cp gpurun_out/r05/tuned_retune_m1.json flute_amd/data/gfx950_tuned.json      # (the box's copy of the tree: the bench below runs on the new table)
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_steps20_run2.json 2> gpurun_out/r05/bench_steps20_run2.err
timeout 300 python bench.py --steps 2000 --warmup 50 > gpurun_out/r05/bench_steps2000_run2.json 2> gpurun_out/r05/bench_steps2000_run2.err
python - <<'PY'
import json
for f in ("gpurun_out/r05/bench_steps20_run2.json", "gpurun_out/r05/bench_steps2000_run2.json"):
    try:
        d = json.loads(open(f).readline())
        print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel_us"], d["roofline"]["kernel_us_hip_events"], d["config"]["template_id"], d["config"]["plan"]["one_shot"], d.get("m256", {}).get("us"), d["eager_us_per_step"])
        for e in d["extras"]: print("   ", e["workload"][:70], e["us"], e.get("template_id"))
    except Exception as e:
        print(f, "ERR", e)
PY
timeout 600 bash tools/prof_bench.sh > gpurun_out/r05/prof_bench.log 2>&1
tail -30 gpurun_out/r05/prof_bench.log
}

call8() {
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "lean or persistent or full_size or decode_plan or fuzz or golden or forced" > gpurun_out/r05/pytest_lean4.log 2>&1
tail -4 gpurun_out/r05/pytest_lean4.log
rm -f gpurun_out/r05/time_cases_rows.jsonl
C=""
for s in "4096,4096" "11008,4096" "8192,4096" "6144,4096" "4096,2048"; do
  for m in 2 3 4; do
    C="$C;4,$m,$s,f16;4,$m,$s,f16,family=0,one_shot=4,waves=4;4,$m,$s,f16,family=0,one_shot=4,waves=8"
  done
done
C="$C;4,1,28672,8192,f16;4,2,28672,8192,f16;4,1,8192,28672,f16;4,1,14336,4096,f16;4,1,28672,4096,f16;4,1,8192,8192,f16;4,1,4096,4096,f16;4,1,11008,4096,f16"
timeout 900 python tools/time_cases.py "${C:1}" --steps 300 --tag rows --out gpurun_out/r05/time_cases_rows.jsonl > gpurun_out/r05/time_cases_rows.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r05/time_cases_rows.jsonl"):
    d = json.loads(l)
    print(d["M"], d["N"], d["K"], d["ovr"], d["tid"], d["us"], "fam", d["plan"]["family"], "os", d["plan"]["one_shot"], d["plan"]["waves"], d["plan"]["kw"], d["plan"]["grid"])
PY
grep -c error gpurun_out/r05/time_cases_rows.log
}

call9() {
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
cp flute_amd/data/gfx950_tuned.json gpurun_out/r05/tuned_retune_m2_m4.json
timeout 700 python -m flute_amd.tune --shapes "1024,4096;3584,4096;4096,4096;4608,4096;6144,4096;8192,4096;11008,4096;14336,4096;16384,4096;28672,4096;4608,2048;8192,2048;16384,2048" \
    --ms 2,4 --bits 4 --groups 64,128 --retune --rep 120 --budget-s 600 --out gpurun_out/r05/tuned_retune_m2_m4.json > gpurun_out/r05/retune_m2_m4.log 2>&1
tail -1 gpurun_out/r05/retune_m2_m4.log
cp gpurun_out/r05/tuned_retune_m2_m4.json flute_amd/data/gfx950_tuned.json
timeout 1100 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r05/pytest_gpu_run2.log 2>&1
tail -14 gpurun_out/r05/pytest_gpu_run2.log
}

call10() {
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "lean_mfma" > gpurun_out/r05/pytest_fastm.log 2>&1
tail -15 gpurun_out/r05/pytest_fastm.log | cut -c1-250
rm -f gpurun_out/r05/time_cases_fastm.jsonl
C=""
for s in "4096,4096" "4096,2048" "8192,4096" "11008,4096" "2048,4096"; do
  for m in 3 4 5 8 16; do
    C="$C;4,$m,$s,f16;4,$m,$s,f16,family=7"
  done
done
C="$C;4,16,4096,4096,bf16;4,16,4096,4096,bf16,family=7"
timeout 900 python tools/time_cases.py "${C:1}" --steps 300 --tag fastm --out gpurun_out/r05/time_cases_fastm.jsonl > gpurun_out/r05/time_cases_fastm.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r05/time_cases_fastm.jsonl"):
    d = json.loads(l)
    print(d["M"], d["N"], d["K"], d["dtype"], d["ovr"], d["tid"], d["us"], "fam", d["plan"]["family"], "os", d["plan"]["one_shot"], d["plan"]["waves"], d["plan"]["grid"])
PY
grep -c error gpurun_out/r05/time_cases_fastm.log
}

call11() {
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
FLUTE_AMD_LIB=$PWD/flute_amd/csrc/libflute_amd_stamps.so timeout 300 python tools/stamps_fast.py "16,4096,4096,family=7;5,4096,4096,family=7;16,4096,2048,family=7;1,4096,4096;4,4096,4096;16,11008,4096,family=7" > gpurun_out/r05/stamps_fastm_run1.jsonl 2> gpurun_out/r05/stamps_fastm_run1.err
cat gpurun_out/r05/stamps_fastm_run1.jsonl
tail -3 gpurun_out/r05/stamps_fastm_run1.err
}

call12() {
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "lean_mfma" > gpurun_out/r05/pytest_fastm2.log 2>&1
tail -5 gpurun_out/r05/pytest_fastm2.log | cut -c1-250
rm -f gpurun_out/r05/time_cases_fastm2.jsonl
C=""
for s in "4096,4096" "4096,2048" "2048,4096" "8192,4096"; do
  for m in 5 8 16; do
    C="$C;4,$m,$s,f16;4,$m,$s,f16,family=7"
  done
done
timeout 900 python tools/time_cases.py "${C:1}" --steps 300 --tag fastm2 --out gpurun_out/r05/time_cases_fastm2.jsonl > gpurun_out/r05/time_cases_fastm2.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r05/time_cases_fastm2.jsonl"):
    d = json.loads(l)
    print(d["M"], d["N"], d["K"], d["dtype"], d["ovr"], d["tid"], d["us"], "fam", d["plan"]["family"], d["plan"]["waves"], d["plan"]["grid"])
PY
FLUTE_AMD_LIB=$PWD/flute_amd/csrc/libflute_amd_stamps.so timeout 300 python tools/stamps_fast.py "16,4096,4096,family=7;5,4096,4096,family=7;16,4096,2048,family=7" > gpurun_out/r05/stamps_fastm_run2.jsonl 2> gpurun_out/r05/stamps_fastm_run2.err
cat gpurun_out/r05/stamps_fastm_run2.jsonl
}

call13() {
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "lean_mfma" > gpurun_out/r05/pytest_fastm3.log 2>&1
tail -5 gpurun_out/r05/pytest_fastm3.log | cut -c1-250
rm -f gpurun_out/r05/time_cases_fastm3.jsonl
C=""
for s in "4096,4096" "4096,2048" "2048,4096"; do
  for m in 5 8 16; do
    C="$C;4,$m,$s,f16,family=7,waves=8;4,$m,$s,f16,family=7,waves=12"
  done
done
timeout 900 python tools/time_cases.py "${C:1}" --steps 300 --tag fastm3 --out gpurun_out/r05/time_cases_fastm3.jsonl > gpurun_out/r05/time_cases_fastm3.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r05/time_cases_fastm3.jsonl"):
    d = json.loads(l)
    print(d["M"], d["N"], d["K"], d["dtype"], d["ovr"], d["tid"], d["us"], "fam", d["plan"]["family"], d["plan"]["waves"], d["plan"]["grid"])
PY
FLUTE_AMD_LIB=$PWD/flute_amd/csrc/libflute_amd_stamps.so timeout 300 python tools/stamps_fast.py "16,4096,4096,family=7,waves=12;5,4096,4096,family=7,waves=12" > gpurun_out/r05/stamps_fastm_run3.jsonl 2> gpurun_out/r05/stamps_fastm_run3.err
cat gpurun_out/r05/stamps_fastm_run3.jsonl
}

call14() {
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
cp flute_amd/data/gfx950_tuned.json gpurun_out/r05/tuned_retune_m16.json
timeout 300 python -m flute_amd.tune --shapes "3584,4096;4096,4096;2048,4096;4096,2048" --ms 16 --bits 4 --groups 64,128 --retune --rep 120 --budget-s 250 --out gpurun_out/r05/tuned_retune_m16.json > gpurun_out/r05/retune_m16.log 2>&1
tail -1 gpurun_out/r05/retune_m16.log
cp gpurun_out/r05/tuned_retune_m16.json flute_amd/data/gfx950_tuned.json
timeout 1100 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/r05/pytest_gpu_run3.log 2>&1
tail -9 gpurun_out/r05/pytest_gpu_run3.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_steps20_run3.json 2> gpurun_out/r05/bench_steps20_run3.err
timeout 300 python bench.py --steps 2000 --warmup 50 > gpurun_out/r05/bench_steps2000_run3.json 2> gpurun_out/r05/bench_steps2000_run3.err
python - <<'PY'
import json
for f in ("gpurun_out/r05/bench_steps20_run3.json", "gpurun_out/r05/bench_steps2000_run3.json"):
    try:
        d = json.loads(open(f).readline())
        print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel_us_hip_events"], d["config"]["template_id"], d["config"]["plan"]["one_shot"], d["m256"]["us"] if d.get("m256") else None, d["eager_us_per_step"])
        for e in d["extras"]: print("   ", e["workload"][:75], e["us"], e.get("template_id"), e.get("speedup_vs_torch_mm"))
        print("   tp", d.get("tp_mlp_pair", {}).get("kernels_us"))
    except Exception as e:
        print(f, "ERR", e)
PY
}

call15() {
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
timeout 1100 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/r05/pytest_gpu_run4.log 2>&1
tail -12 gpurun_out/r05/pytest_gpu_run4.log | cut -c1-300
rm -f gpurun_out/r05/time_cases_w3.jsonl
timeout 600 python tools/time_cases.py "3,1,8192,8192,bf16;3,1,28672,8192,bf16;3,1,8192,28672,bf16;3,2,28672,8192,bf16;3,1,4096,4096,bf16;3,1,14336,4096,f16;4,1,4096,8192,f16;4,1,8192,8192,f16;4,2,8192,8192,f16;4,4,8192,4096,f16,family=0;2,1,8192,8192,f16" --steps 300 --tag w3 --out gpurun_out/r05/time_cases_w3.jsonl > gpurun_out/r05/time_cases_w3.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r05/time_cases_w3.jsonl"):
    d = json.loads(l)
    print(d["bits"], d["M"], d["N"], d["K"], d["dtype"], d["ovr"], d["tid"], d["us"], "fam", d["plan"]["family"], "os", d["plan"]["one_shot"], d["plan"]["waves"], d["plan"]["grid"])
PY
}

call16() {
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
timeout 600 bash tools/prof_bench.sh > gpurun_out/r05/prof_bench2.log 2>&1
tail -12 gpurun_out/r05/prof_bench2.log | cut -c1-250
grep "^PMC\|^TRACE" gpurun_out/prof/bench/summary.txt | grep -i "qgem" | cut -c1-400
timeout 600 bash tools/prof_m256.sh > gpurun_out/r05/prof_m256.log 2>&1
tail -3 gpurun_out/r05/prof_m256.log | cut -c1-1200
grep "^PMC\|^TRACE" gpurun_out/prof/m256/summary.txt | grep -i "qgem\|splitk" | cut -c1-400
timeout 400 bash tools/prof_case.sh fastm_m16 --M 16 --N 4096 --K 4096 --tid 16 --steps 60 > gpurun_out/r05/prof_fastm.log 2>&1
grep "^PMC\|^TRACE" gpurun_out/prof/fastm_m16/summary.txt | grep -i "qgem" | cut -c1-400
}

call17() {
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_qgemm_gpu.py tests/test_parity_sweep_gpu.py -x -q -m gpu -k "lean or hadamard or higgs or golden" > gpurun_out/r05/pytest_had.log 2>&1
tail -8 gpurun_out/r05/pytest_had.log | cut -c1-300
rm -f gpurun_out/r05/time_cases_k3584.jsonl
timeout 600 python tools/time_cases.py "4,1,4096,3584,f16;4,1,4096,3584,f16,one_shot=1;4,1,4096,3584,f16,one_shot=3;4,2,4096,3584,f16;4,2,4096,3584,f16,one_shot=1;4,4,4096,3584,f16;4,4,4096,3584,f16,family=0,one_shot=1;4,1,2048,3584,f16;4,1,8192,3584,f16;4,1,8192,3584,f16,one_shot=4;4,1,14336,3584,f16;4,1,14336,3584,f16,one_shot=4" --steps 300 --tag k3584 --out gpurun_out/r05/time_cases_k3584.jsonl > gpurun_out/r05/time_cases_k3584.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r05/time_cases_k3584.jsonl"):
    d = json.loads(l)
    print(d["bits"], d["M"], d["N"], d["K"], d["dtype"], d["ovr"], d["tid"], d["us"], "fam", d["plan"]["family"], "os", d["plan"]["one_shot"], d["plan"]["waves"], d["plan"]["grid"])
PY
python - <<'PY'
# configs[4]: pair codebook + Hadamard 512 on 3584 x 4096 (N = 4096, K = 3584), M = 1: the bench's line
import torch, bench
d = torch.device("cuda:0")
lay = bench.Layer(1, 4096, 3584, 4, 64, torch.float16, d, bench.copies_for(4096, 3584, 4), None, hadamard_size=512)
lay.tune()
us = min(bench.time_graph(lay, 300, 20, torch.cuda.synchronize)[0] for _ in range(3)) / 300 * 1e3
from flute_amd import utils
print("configs[4] 3584x4096 hadamard 512 M=1:", round(us, 3), "us tid", lay.template_id)
lay = bench.Layer(1, 3584, 4096, 4, 64, torch.float16, d, bench.copies_for(3584, 4096, 4), None, hadamard_size=512)
lay.tune()
us = min(bench.time_graph(lay, 300, 20, torch.cuda.synchronize)[0] for _ in range(3)) / 300 * 1e3
print("4096x3584 (N=3584, K=4096) hadamard 512 M=1:", round(us, 3), "us tid", lay.template_id)
PY
}

call18() {
set -u
cd "$(dirname "$0")/.."
python - <<'PY'
import torch, bench
from flute_amd import dev, utils
d = torch.device("cuda:0")
for (N, K) in ((4096, 3584), (3584, 4096)):
    for M in (1, 2):
        if M * K > 8192: continue
        for tid in (16, 17, 1, 0, 18):
            lay = bench.Layer(M, N, K, 4, 64, torch.float16, d, bench.copies_for(N, K, 4), None, hadamard_size=512)
            lay.template_id = tid
            us = min(bench.time_graph(lay, 300, 20, torch.cuda.synchronize)[0] for _ in range(3)) / 300 * 1e3
            print("had512", M, N, K, "tid", tid, round(us, 3), flush=True)
            del lay; torch.cuda.empty_cache()
        lay = bench.Layer(M, N, K, 4, 64, torch.float16, d, bench.copies_for(N, K, 4), None)
        lay.template_id = 16
        us = min(bench.time_graph(lay, 300, 20, torch.cuda.synchronize)[0] for _ in range(3)) / 300 * 1e3
        print("plain ", M, N, K, "tid 16", round(us, 3), flush=True)
        del lay; torch.cuda.empty_cache()
PY
}

call19() {
# round 5, GPU calls 19 and 20: the lane-stage rewrite of the transform (DPP + lane swaps instead of ds_bpermute): tests, then the
# Hadamard timings of call 18 again (lean kernel with the fused rotation: ids 16/0; round-4 one-shot kernel: ids 17/1)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "hadamard or higgs or lean or golden or fwht" 2>&1 | tail -5
python - <<'PY'
import torch, bench
from flute_amd import dev, utils
d = torch.device("cuda:0")
for (N, K) in ((4096, 3584), (3584, 4096), (4096, 4096)):
    for M in (1, 2):
        if M * K > 8192: continue
        for tid in (16, 17, 1, 0):
            lay = bench.Layer(M, N, K, 4, 64, torch.float16, d, bench.copies_for(N, K, 4), None, hadamard_size=512)
            lay.template_id = tid
            us = min(bench.time_graph(lay, 300, 20, torch.cuda.synchronize)[0] for _ in range(3)) / 300 * 1e3
            print("had512", M, N, K, "tid", tid, round(us, 3), flush=True)
            del lay; torch.cuda.empty_cache()
# the stand-alone transform
import flute_amd
for rows, n, h in ((1, 3584, 512), (1, 4096, 4096), (16, 4096, 512), (256, 4096, 4096), (4096, 4096, 4096)):
    x = torch.randn(rows, n, dtype=torch.float16, device=d)
    y = torch.empty_like(x)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        from flute_amd import ops as hm
        for _ in range(3): hm.hadamard_transform(x, h)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(100): hm.hadamard_transform(x, h)
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 10)
    print("fwht", rows, n, h, round(best, 2), "us", round(rows * n * 4 / best / 1e3, 1), "GB/s", flush=True)
PY
}

call21() {
# round 5, GPU call 21: K = 3584 decode keys measured again with the lean kernel's 7-piece shape, the whole GPU suite, bench lines
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m flute_amd.tune --retune --shapes '4096,3584;8192,3584;2048,3584;14336,3584' --ms 1,2,4 --bits 4 2>&1 | tail -5
cp flute_amd/data/gfx950_tuned.json gpurun_out/gfx950_tuned.json
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_20.json 2> gpurun_out/bench_20.err; tail -c 3000 gpurun_out/bench_20.json
python bench.py --steps 2000 --warmup 50 > gpurun_out/bench_2000.json 2> gpurun_out/bench_2000.err; tail -c 1500 gpurun_out/bench_2000.json
}

call22() {
# round 5, GPU call 22: (a) bench's 20-step line with the copies in one arena against separate tensors; (b) the Hadamard
# fuse threshold again now that a rotation costs half of what it did (M = 3, 4, 8: fused by override against the operator's two launches)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for rep in 1 2; do
for a in 1 0; do
  FLUTE_BENCH_ARENA=$a python bench.py --steps 20 --warmup 5 --no-extras --no-cpu 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = j['roofline']
print('arena', $a, 'steps 20', j['value'], r.get('kernel_us'), r.get('kernel_us_hip_events'))"
done
done
FLUTE_BENCH_ARENA=1 python bench.py --steps 2000 --warmup 50 --no-extras --no-cpu 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = j['roofline']
print('arena 1 steps 2000', j['value'], r.get('kernel_us'), r.get('kernel_us_hip_events'))"
python - <<'PY'
import torch, bench
from flute_amd import dev
d = torch.device("cuda:0")
for (N, K) in ((4096, 3584), (4096, 4096), (3584, 4096), (14336, 3584)):
    for M in (2, 3, 4, 8):
        row = []
        for forced in (False, True):
            lay = bench.Layer(M, N, K, 4, 64, torch.float16, d, bench.copies_for(N, K, 4), None, hadamard_size=512)
            lay.template_id = 16
            if forced: lay.ovr = dev.Overrides(family=0)
            try:
                us = min(bench.time_graph(lay, 300, 20, torch.cuda.synchronize)[0] for _ in range(3)) / 300 * 1e3
            except Exception as e:
                us = float("nan")
            row.append(round(us, 3))
            del lay; torch.cuda.empty_cache()
        print("had512 M", M, N, K, "operator", row[0], "forced fused", row[1], flush=True)
PY
}

call23() {
# round 5, GPU call 23: the decode buckets of the tuned table measured again (the kernels behind them changed this round)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1000 python -m flute_amd.tune --retune --shapes supported --ms 1,2,4 --budget-s 800 2>&1 | tail -3
cp flute_amd/data/gfx950_tuned.json gpurun_out/gfx950_tuned.json
}

call24() {
# round 5, GPU call 24: (a) the 3-bit blocks' K slices combined inside the launch (xwg_seam) against the reduce launch;
# (b) the 3-bit block kernel's whole-line PLANE pieces (qgemm_block3.h FLUTE_B3_LINE_PLANES = 1 / 2, written at the end of round 4,
# never run): parity, then the 3-bit prefill / mid-M cases against the default build
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/time_cases.jsonl
MID="3,1024,4096,4096,bf16;3,256,8192,8192,bf16;3,512,4096,4096,bf16;3,512,8192,8192,bf16;3,128,8192,8192,bf16;3,64,8192,8192,bf16;3,256,4096,4096,bf16;3,1024,4096,4096,f16;3,96,28672,8192,bf16"
PRE="3,4096,4096,4096,bf16;3,1024,28672,8192,bf16;3,4096,4096,4096,f16;3,2048,4096,11008,bf16"
echo "== default build: tests"
timeout 600 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "block_prefill or seam_under_load or splitk or fuzz" 2>&1 | tail -3
echo "== in-launch combine"
timeout 300 python tools/time_cases.py "$MID" --tag inlaunch 2>&1 | cut -c1-260
echo "== reduce launch"
FLUTE_AMD_B3_TWO_LAUNCH=1 timeout 300 python tools/time_cases.py "$MID" --tag twolaunch 2>&1 | cut -c1-260
echo "== prefill, default"
timeout 300 python tools/time_cases.py "$PRE" --tag lp0 2>&1 | cut -c1-200
cp flute_amd/csrc/libflute_amd.so /tmp/libflute_amd_default.so
for v in lp1 lp2; do
  cp flute_amd/csrc/libflute_amd_$v.so flute_amd/csrc/libflute_amd.so
  echo "== $v"
  timeout 300 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "block_prefill" 2>&1 | tail -2
  timeout 300 python tools/time_cases.py "$PRE;3,1024,4096,4096,bf16;3,256,8192,8192,bf16" --tag $v 2>&1 | cut -c1-200
done
cp /tmp/libflute_amd_default.so flute_amd/csrc/libflute_amd.so
}

call25() {
# round 5, GPU call 25: (a) the restricted in-launch seam build under test; (b) M = 64 on 8192 x 28672 (round 4's one regret case over
# 10 %) with the split-K block kernel forced; (c) the tuner's challenge pass on the M = 64 bucket (4 bits, ids 0 / 16) and the M = 128
# bucket (2 bits); (d) the regret sweep at the batch sizes between the swept ones and at the swept ones (the kernels changed)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/time_cases.jsonl
timeout 600 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "block_prefill or seam_under_load or splitk" 2>&1 | tail -3
timeout 200 python tools/time_cases.py "4,64,28672,8192,f16;4,64,28672,8192,f16,family=6,m_tiles=4;4,64,28672,8192,f16,family=6,m_tiles=8;4,48,28672,8192,f16;4,48,28672,8192,f16,family=6,m_tiles=4;4,33,28672,8192,f16;4,33,28672,8192,f16,family=6,m_tiles=4" --tag m64 2>&1 | cut -c1-330
cp flute_amd/data/gfx950_tuned.json gpurun_out/tuned_challenged.json
timeout 260 python -m flute_amd.tune --out gpurun_out/tuned_challenged.json --ms 64 --bits 4 --groups 64,128 --challenge 0,16 \
    --rep 20 --budget-s 240 > gpurun_out/challenge_m64_b4.log 2>&1
timeout 160 python -m flute_amd.tune --out gpurun_out/tuned_challenged.json --ms 128 --bits 2 --groups 64 --challenge 0,7,3,15 \
    --rep 20 --budget-s 140 > gpurun_out/challenge_m128_b2.log 2>&1
cp gpurun_out/tuned_challenged.json flute_amd/data/gfx950_tuned.json
timeout 330 python tools/regret.py --ms 32,48,96,128,384,512,2048 --budget-s 300 --steps 60 \
    --out gpurun_out/planner_regret_between.json > gpurun_out/regret_between.log 2>&1
timeout 400 python tools/regret.py --ms 1,2,4,16,64,256,1024 --budget-s 370 --steps 60 \
    --out gpurun_out/planner_regret_r05.json > gpurun_out/regret_r05.log 2>&1
tail -3 gpurun_out/challenge_m64_b4.log gpurun_out/challenge_m128_b2.log gpurun_out/regret_between.log gpurun_out/regret_r05.log
}

call26() {
# round 5, GPU call 26: after the planner fixes from call 25's regret sweeps (3-bit blocks x uneven K slices, cheapest K-split candidate,
# digit 3 = no lane sharing + two slabs per wave above M = 16): the 4-bit M = 32 / 64 / 128 buckets tuned again, both regret sweeps again
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 500 python -m flute_amd.tune --retune --shapes supported --ms 32,64,128 --bits 4 --rep 20 --budget-s 420 2>&1 | tail -2
cp flute_amd/data/gfx950_tuned.json gpurun_out/gfx950_tuned.json
timeout 330 python tools/regret.py --ms 32,48,96,128,384,512,2048 --budget-s 300 --steps 60 \
    --out gpurun_out/planner_regret_between_after.json > gpurun_out/regret_between_after.log 2>&1
timeout 300 python tools/regret.py --ms 1,2,4,16,64,256,1024 --budget-s 270 --steps 60 \
    --out gpurun_out/planner_regret_r05_after.json > gpurun_out/regret_r05_after.log 2>&1
tail -n 1 gpurun_out/regret_between_after.log gpurun_out/regret_r05_after.log
timeout 300 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "block_prefill or seam_under_load or mfma_family or fuzz" 2>&1 | tail -2
}

call27() {
# round 5, GPU call 27: after the deep-layer rule (grid K split instead of more lane sharing) and the skinny-block fill rule: the buckets
# whose automatic ids changed plans tuned again, the regret sweep of the swept batch sizes and of the ones between
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 400 python -m flute_amd.tune --retune --shapes supported --ms 4,16,32 --bits 4,2 --rep 20 --budget-s 330 2>&1 | tail -1
cp flute_amd/data/gfx950_tuned.json gpurun_out/gfx950_tuned.json
timeout 200 python tools/regret.py --ms 1,2,4,16,64,256,1024 --budget-s 170 --steps 60 \
    --out gpurun_out/planner_regret_r05_final.json > gpurun_out/regret_r05_final.log 2>&1
timeout 260 python tools/regret.py --ms 32,48,96,128,384,512,2048 --budget-s 240 --steps 60 \
    --out gpurun_out/planner_regret_between_final.json > gpurun_out/regret_between_final.log 2>&1
tail -n 1 gpurun_out/regret_r05_final.log gpurun_out/regret_between_final.log
}

call28() {
# round 5, GPU call 28: the split-K block kernel's K-split launches in XCD-group order (the row tiles of a (column tile, slice) on one XCD)
# against the natural order: parity / seam tests, then timings; traffic of M = 256 on 4096^2 (PMC) with the new order
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/time_cases.jsonl
timeout 600 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "splitk or seam_under_load or fuzz" 2>&1 | tail -2
CASES="4,256,4096,4096,f16;4,256,4096,4096,bf16;4,128,4096,4096,f16;4,64,8192,8192,f16;4,96,14336,4096,f16;4,256,8192,8192,f16;4,512,4096,4096,f16;4,80,8192,8192,f16;2,96,8192,8192,f16;4,1024,4096,4096,f16;4,384,4096,4096,f16"
echo "== xcd-group order"
timeout 300 python tools/time_cases.py "$CASES" --tag xcdgroups 2>&1 | cut -c1-290
echo "== natural order"
FLUTE_AMD_SK_NATURAL=1 timeout 300 python tools/time_cases.py "$CASES" --tag natural 2>&1 | cut -c1-290
}

call29() {
# round 5, GPU calls 29 and 36 (final library): the whole GPU suite, smoke(), the bench lines (driver's 20 steps and 2 000), rocprofv3 passes of the bench
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_final.log 2>&1; tail -3 gpurun_out/pytest_gpu_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_20.json 2> gpurun_out/bench_20.err
python bench.py --steps 2000 --warmup 50 > gpurun_out/bench_2000.json 2> gpurun_out/bench_2000.err
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
for f in bench_20 bench_2000 bench_default; do python - $f <<'PY'
import json, sys
j = json.loads(open("gpurun_out/" + sys.argv[1] + ".json").read().strip().splitlines()[-1]); r = j["roofline"]
print(sys.argv[1], j["value"], j["unit"], "us", r.get("kernel_us"), "events", r.get("kernel_us_hip_events"), "m256", j.get("m256", {}).get("us"))
PY
done
}

call30() {
# round 5, GPU call 30: randomised parity sweeps on the final library (automatic plans over random template ids; the lean kernels' shapes)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
FUZZ_OUT=gpurun_out/r05_gpu_fuzz.json timeout 500 python tools/gpu_fuzz.py 24000 505 2>&1 | tail -4
FUZZ_DECODE=3 FUZZ_OUT=gpurun_out/r05_gpu_fuzz_lean_shapes.json timeout 300 python tools/gpu_fuzz.py 16000 506 2>&1 | tail -4
}

call31() {
# round 5, GPU call 31: M = 256 on 4096^2 - forced variants of the split-K block kernel (loader waves off, bf16, 128-row tiles) next to the automatic plan
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/time_cases.jsonl
timeout 400 python tools/time_cases.py "4,256,4096,4096,f16;4,256,4096,4096,f16,family=6,m_tiles=4,splitk=2;4,256,4096,4096,f16,family=6,m_tiles=4,splitk=2,waves=8;4,256,4096,4096,f16,family=6,m_tiles=8,splitk=4;4,256,4096,4096,f16,family=6,m_tiles=4,splitk=4;4,256,4096,4096,f16,family=2;4,256,4096,4096,bf16;4,256,4096,4096,bf16,family=6,m_tiles=4,splitk=2;4,256,4096,4096,bf16,family=6,m_tiles=4,splitk=2,waves=8;4,200,4096,4096,f16;4,200,4096,4096,f16,family=6,m_tiles=4,splitk=2;4,192,4096,4096,f16;4,192,4096,4096,f16,family=6,m_tiles=4,splitk=2" --steps 400 --tag m256 2>&1 | cut -c1-300
}

call32() {
# round 5, GPU call 32: two / four rows on K = 4096 layers of more than one round of workgroups: the lean kernel forced (one_shot = 4) against the automatic plan
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/time_cases.jsonl
C=""
for N in 8192 11008 14336; do for M in 2 3 4; do C="$C;4,$M,$N,4096,f16;4,$M,$N,4096,f16,family=0,one_shot=4"; done; done
for N in 6144 8192; do for M in 2 4; do C="$C;4,$M,$N,2048,f16;4,$M,$N,2048,f16,family=0,one_shot=4"; done; done
timeout 500 python tools/time_cases.py "${C:1}" --steps 300 --tag rows 2>&1 | cut -c1-250
}

call33() {
# round 5, GPU call 33: split-K block kernel with the per-half-step preparation (scale multiplies, next scale reads, word shuffle) moved in front of the barrier
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/time_cases.jsonl
timeout 600 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "splitk or seam_under_load" 2>&1 | tail -2
timeout 400 python tools/time_cases.py "4,256,4096,4096,f16;4,256,4096,4096,bf16,family=6,m_tiles=4,splitk=2;4,256,11008,4096,f16;4,1024,4096,4096,f16;4,64,8192,8192,f16;4,96,14336,4096,f16;4,256,8192,8192,f16;4,512,4096,4096,f16;2,96,8192,8192,f16;4,128,8192,8192,f16" --steps 400 --tag prep_before_barrier 2>&1 | cut -c1-250
}

call34() {
# round 5, GPU call 34: A/B on one box - split-K block kernel with the per-half-step preparation in front of the barrier (new) against the committed kernel (old), alternating
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/time_cases.jsonl
C="4,256,4096,4096,f16;4,256,11008,4096,f16;4,1024,4096,4096,f16;4,64,8192,8192,f16;4,96,14336,4096,f16;4,256,8192,8192,f16;4,512,4096,4096,f16;4,128,8192,8192,f16"
cp flute_amd/csrc/libflute_amd.so /tmp/new.so
for rep in 1 2; do
  for v in old new; do
    if [ $v = old ]; then cp flute_amd/csrc/libflute_amd_oldsk.so flute_amd/csrc/libflute_amd.so; else cp /tmp/new.so flute_amd/csrc/libflute_amd.so; fi
    timeout 300 python tools/time_cases.py "$C" --steps 400 --tag $v$rep 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['tag'], j['M'], j['N'], j['K'], j['us'], j['plan']['m_tiles'], j['plan']['splitk'])"
  done
done
cp /tmp/new.so flute_amd/csrc/libflute_amd.so
}

call35() {
# round 5, GPU call 35: A/B on one box - block prefill kernels (qgemm_block2.h / qgemm_block3.h) with the half step's register-only preparation in
# front of the barrier (new) against the committed kernels (old), alternating; parity of the new ones first
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/time_cases.jsonl
timeout 600 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "block_prefill or seam_under_load or full_size" 2>&1 | tail -2
C="4,4096,4096,4096,f16;4,2048,4096,4096,f16;4,4096,4096,4096,bf16;4,2048,4096,4096,bf16;4,4096,11008,4096,f16;3,4096,4096,4096,bf16;3,1024,4096,4096,bf16;2,4096,4096,4096,f16;3,1024,28672,8192,bf16"
cp flute_amd/csrc/libflute_amd.so /tmp/new.so
for rep in 1 2; do
  for v in old new; do
    if [ $v = old ]; then cp flute_amd/csrc/libflute_amd_oldblk.so flute_amd/csrc/libflute_amd.so; else cp /tmp/new.so flute_amd/csrc/libflute_amd.so; fi
    timeout 300 python tools/time_cases.py "$C" --steps 100 --tag $v$rep 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['tag'], j['bits'], j['M'], j['N'], j['K'], j['dtype'], j['us'], j['plan']['family'], j['plan']['m_block'], j['plan']['splitk'])"
  done
done
cp /tmp/new.so flute_amd/csrc/libflute_amd.so
}

call37() {
# round 5, GPU call 37: split-K block kernel, loader-wave variant: stage hand-off through LDS words instead of two workgroup barriers per step (new)
# against the committed kernel (old), alternating on one box; parity of the new one first (its own timeout: a hand-off bug would hang)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/time_cases.jsonl
timeout 240 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "splitk or seam_under_load" 2>&1 | tail -3
C="4,256,4096,4096,f16;4,64,8192,8192,f16;4,128,8192,8192,f16;4,96,8192,8192,f16;4,1024,4096,4096,bf16;4,256,11008,4096,bf16,family=6,m_tiles=8,splitk=1;4,512,4096,4096,bf16,family=6,m_tiles=8,splitk=2"
cp flute_amd/csrc/libflute_amd.so /tmp/new.so
for rep in 1 2; do
  for v in old new; do
    if [ $v = old ]; then cp flute_amd/csrc/libflute_amd_oldsk.so flute_amd/csrc/libflute_amd.so; else cp /tmp/new.so flute_amd/csrc/libflute_amd.so; fi
    timeout 200 python tools/time_cases.py "$C" --steps 300 --tag $v$rep 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['tag'], j['bits'], j['M'], j['N'], j['K'], j['dtype'], j['us'], j['plan']['family'], j['plan']['m_tiles'], j['plan']['splitk'])"
  done
done
cp /tmp/new.so flute_amd/csrc/libflute_amd.so
}

call38() {
# round 5, GPU call 38: rocprofv3 trace + PMC passes of two kernels this round changed: the 3-bit 128-row blocks x 2 K slices combined in the launch
# (M = 1024 on 4096^2, bf16) and the fused-rotation decode of configs[4] (3584 x 4096 [N x K], Hadamard 512)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 500 bash tools/prof_case.sh w3_m1024_4096_inlaunch --bits 3 --M 1024 --N 4096 --K 4096 --bf16 --tid 4 2>&1 | tail -6
timeout 400 bash tools/prof_case.sh lean_m1_4096x3584 --bits 4 --M 1 --N 4096 --K 3584 2>&1 | tail -4
}

call39() {
# round 5, GPU calls 39 and 41: bf16 scale multiply of the MFMA kernels as two v_dot2_f32_bf16 + one v_cvt_pk_bf16_f32 per word (new) against unpack / multiply / convert
# (old), alternating on one box; parity of the new one first (one-hot rows bit-exact against round_bf16(lut * s))
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/time_cases.jsonl
timeout 900 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "block_prefill or splitk or mfma or skinny or golden or random_vs_oracle or group_sizes or full_size or fuzz or seam" 2>&1 | tail -4
C="4,4096,4096,4096,bf16;4,2048,4096,4096,bf16;4,1024,4096,4096,bf16;4,256,11008,4096,bf16;3,4096,4096,4096,bf16;3,1024,4096,4096,bf16;3,256,8192,8192,bf16;3,1024,28672,8192,bf16;4,64,8192,8192,bf16;4,16,28672,8192,bf16"
cp flute_amd/csrc/libflute_amd.so /tmp/new.so
for rep in 1 2; do
  for v in old new; do
    if [ $v = old ]; then cp flute_amd/csrc/libflute_amd_oldbf.so flute_amd/csrc/libflute_amd.so; else cp /tmp/new.so flute_amd/csrc/libflute_amd.so; fi
    timeout 300 python tools/time_cases.py "$C" --steps 100 --tag $v$rep 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['tag'], j['bits'], j['M'], j['N'], j['K'], j['dtype'], j['us'], j['plan']['family'], j['plan']['m_block'], j['plan']['splitk'])"
  done
done
cp /tmp/new.so flute_amd/csrc/libflute_amd.so
}

call42() {
# round 5, GPU call 42 (final library): the whole GPU suite, smoke(), fuzz sweeps, the bench lines
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_final.log 2>&1; tail -3 gpurun_out/pytest_gpu_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
FUZZ_OUT=gpurun_out/r05_gpu_fuzz.json timeout 500 python tools/gpu_fuzz.py 24000 505 2>&1 | grep "^fuzz"
FUZZ_DECODE=3 FUZZ_OUT=gpurun_out/r05_gpu_fuzz_lean_shapes.json timeout 300 python tools/gpu_fuzz.py 16000 506 2>&1 | grep "^fuzz"
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_20.json 2> gpurun_out/bench_20.err
python bench.py --steps 2000 --warmup 50 > gpurun_out/bench_2000.json 2> gpurun_out/bench_2000.err
for f in bench_20 bench_2000; do python - $f <<'PY'
import json, sys
j = json.loads(open("gpurun_out/" + sys.argv[1] + ".json").read().strip().splitlines()[-1]); r = j["roofline"]
print(sys.argv[1], j["value"], j["unit"], "us", r.get("kernel_us"), "events", r.get("kernel_us_hip_events"), "m256", j.get("m256", {}).get("us"))
PY
done
}

call43() {
# round 5, GPU call 43: the per-wave MFMA kernel and the skinny kernel with the bf16 scale multiply of common.h's mul_scale4 (new) against the committed
# library (old), alternating on one box; parity first
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/time_cases.jsonl
timeout 900 python -m pytest tests/test_qgemm_gpu.py -x -q -m gpu -k "mfma or skinny or golden or random_vs_oracle or group_sizes or ragged or fuzz" 2>&1 | tail -3
C="4,256,4096,4096,bf16;4,16,28672,8192,bf16;4,64,28672,8192,bf16;4,32,4096,4096,bf16;4,128,4096,4096,bf16;4,16,11008,4096,bf16;3,64,4096,4096,bf16;3,32,8192,8192,bf16;2,64,8192,8192,bf16;4,8,14336,4096,bf16"
cp flute_amd/csrc/libflute_amd.so /tmp/new.so
for rep in 1 2; do
  for v in old new; do
    if [ $v = old ]; then cp flute_amd/csrc/libflute_amd_oldtile.so flute_amd/csrc/libflute_amd.so; else cp /tmp/new.so flute_amd/csrc/libflute_amd.so; fi
    timeout 300 python tools/time_cases.py "$C" --steps 200 --tag $v$rep 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['tag'], j['bits'], j['M'], j['N'], j['K'], j['dtype'], j['us'], j['plan']['family'], j['plan']['m_block'], j['plan']['splitk'])"
  done
done
cp /tmp/new.so flute_amd/csrc/libflute_amd.so
}

call45() {
# round 5, GPU call 45: K = 8192 layers of one round of workgroups (the TP-8 shard of configs[3] among them): the lean kernel's (8, 2, 8) shape forced against the automatic plan
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/time_cases.jsonl
C=""
for N in 3584 4096 2048 1024; do for M in 1 2; do C="$C;4,$M,$N,8192,f16;4,$M,$N,8192,f16,family=0,one_shot=4"; done; done
timeout 300 python tools/time_cases.py "${C:1}" --steps 400 --tag k8192 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['M'], j['N'], j['K'], j['ovr'] or 'auto', j['tid'], j['us'], j['plan']['one_shot'], j['plan']['waves'], j['plan']['kw'], j['plan']['grid'])"
}

call46() {
# round 5, GPU call 46: the lean kernel's K = 8192 shape for two rows (automatic on layers of >= 80 % of a round): parity, the keys concerned tuned again, timing
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/time_cases.jsonl
timeout 600 python -m pytest tests/test_qgemm_gpu.py tests/test_parity_sweep_gpu.py -x -q -m gpu -k "lean or decode_plan or 3584 or fuzz or golden" 2>&1 | tail -2
timeout 200 python -m flute_amd.tune --retune --shapes '3584,8192;4096,8192' --ms 2 --bits 4 2>&1 | tail -1
cp flute_amd/data/gfx950_tuned.json gpurun_out/gfx950_tuned.json
timeout 200 python tools/time_cases.py "4,2,3584,8192,f16;4,2,3584,8192,bf16;4,2,4096,8192,f16;4,1,3584,8192,f16" --steps 400 --tag k8192_auto 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['M'], j['N'], j['K'], j['dtype'], j['tid'], j['us'], j['plan']['one_shot'], j['plan']['waves'], j['plan']['kw'], j['plan']['grid'])"
}

call47() {
# round 5, GPU call 47 (final library): K = 2048 two-row keys tuned again (the lean kernel up to two rounds), then the whole GPU suite, smoke(), the bench lines
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/time_cases.jsonl
timeout 200 python -m flute_amd.tune --retune --shapes '8192,2048;4608,2048;16384,2048' --ms 2 --bits 4 2>&1 | tail -1
cp flute_amd/data/gfx950_tuned.json gpurun_out/gfx950_tuned.json
timeout 200 python tools/time_cases.py "4,2,8192,2048,f16;4,2,4608,2048,f16;4,2,6144,2048,f16;4,2,16384,2048,f16" --steps 400 --tag k2048 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['M'], j['N'], j['K'], j['dtype'], j['tid'], j['us'], j['plan']['one_shot'], j['plan']['waves'], j['plan']['kw'], j['plan']['grid'])"
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_final.log 2>&1; tail -3 gpurun_out/pytest_gpu_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_20.json 2> gpurun_out/bench_20.err
python bench.py --steps 2000 --warmup 50 > gpurun_out/bench_2000.json 2> gpurun_out/bench_2000.err
for f in bench_20 bench_2000; do python - $f <<'PY'
import json, sys
j = json.loads(open("gpurun_out/" + sys.argv[1] + ".json").read().strip().splitlines()[-1]); r = j["roofline"]
print(sys.argv[1], j["value"], j["unit"], "us", r.get("kernel_us"), "events", r.get("kernel_us_hip_events"), "m256", j.get("m256", {}).get("us"))
PY
done
}

"$@"
