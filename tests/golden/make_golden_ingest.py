"""Fixture for the HU -> density ingest (reference diffdrr/data.py:214-227).  data.py cannot be
imported here (nibabel / torchio are absent), so the UNMODIFIED source text of
`transform_hu_to_density` is read from the reference checkout and executed as is.
    python tests/golden/make_golden_ingest.py   ->  tests/golden/hu_to_density.npz"""
import ast
import os

import numpy as np
import torch

REF = os.environ.get("DIFFDRR_REFERENCE", "/root/reference")
src = open(os.path.join(REF, "diffdrr", "data.py")).read()
fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "transform_hu_to_density")
ns = {"torch": torch}
exec(compile(ast.Module(body=[fn], type_ignores=[]), "data.py", "exec"), ns)  # noqa: S102
g = torch.Generator().manual_seed(0)
# int16 CT-like values: air, soft tissue, bone, and values on the two thresholds
vol = (torch.rand(12, 10, 9, generator=g) * 3000 - 1100).round().to(torch.int16)
vol[0, 0, :4] = torch.tensor([-800, -799, 350, 351], dtype=torch.int16)
out = {"volume": vol.numpy()}
for m in (1.0, 2.5):
    out[f"density_{m}"] = ns["transform_hu_to_density"](vol, m).numpy()
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "hu_to_density.npz"), **out)
print({k: v.shape for k, v in out.items()})
