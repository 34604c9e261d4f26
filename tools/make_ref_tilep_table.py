"""Generates flute_amd/data/ref_packed_tilep.json from the reference checkout (build container only).

Hub checkpoints of the reference are packed with the template id its bundled table
`flute/data/qgemm_kernel_raw_tuned_configs.no-M.pth` assigns to (num_sms, num_bits, group_size, N, K, dtype)
(flute/integrations/huggingface.py:53-83); safetensors files do not carry that id.  All the gfx950 loader
needs of it is the packed layout parameter TileP (the native unpacker reads the fields directly), so only
key -> TileP is shipped: wire-format metadata, like tests/golden/ref_template_tilep.json.

    python tools/make_ref_tilep_table.py        # needs /root/reference
"""
import json
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/flute/data"
tuned = torch.load(os.path.join(REF, "qgemm_kernel_raw_tuned_configs.no-M.pth"), weights_only=True)
configs = torch.load(os.path.join(REF, "qgemm_kernel_raw_generated_configs.pth"), weights_only=True)
tile_ps = {}
shapes = set()
for (num_sms, bits, g, N, K, dtype), tid in sorted(tuned.items()):
    tile_ps.setdefault(int(configs[(bits, int(tid))]["TileP"]), 0)
    tile_ps[int(configs[(bits, int(tid))]["TileP"])] += 1
    shapes.add((N, K))
# every id the reference ever tuned (A100 108 SMs, A6000 84, RTX 4090 128; 2/3/4 bits; g 32..256; fp16/bf16) is a
# TileP = 32 template: a published checkpoint without extra state is a TileP-32 layout
assert set(tile_ps) == {32}, tile_ps
path = os.path.join(ROOT, "flute_amd", "data", "ref_packed_tilep.json")
json.dump({"source": "flute v0.4.2 data/qgemm_kernel_raw_tuned_configs.no-M.pth (template id -> TileP via "
                     "data/qgemm_kernel_raw_generated_configs.pth)",
           "entries": len(tuned), "num_sms": sorted({k[0] for k in tuned}), "tile_p_counts": tile_ps,
           "default_tile_p": 32, "shapes": sorted(shapes)}, open(path, "w"), indent=0)
print(path, os.path.getsize(path), "bytes,", len(tuned), "entries, TileP counts", tile_ps)
