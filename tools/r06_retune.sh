#!/bin/bash
# Round 6: the split-K block kernel's candidate list changed (four K parts per workgroup), so the Stages-5 ids of the shipped table
# (family 6, rank 0 / 1 / 2 of the planner's list) and the automatic ids mean other plans at 33 <= M <= 1024: time each key's entry
# against the ids that force the family and the automatic ids, on the GPU box; the table travels through gpurun_out/.
cp flute_amd/data/gfx950_tuned.json gpurun_out/gfx950_tuned.json
python -m flute_amd.tune --shapes supported --ms "$1" --bits "$2" --groups 64,128 --dtypes float16,bfloat16 --out gpurun_out/gfx950_tuned.json \
    --challenge "$3" --rep 30 --budget-s "$4"
