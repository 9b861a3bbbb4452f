"""The reference's example label map as a committed fixture (build container only).

    python tests/golden/make_golden_mask.py

Reads /root/reference/diffdrr/data/mask.nii.gz (TotalSegmentator labels of the example CT the
reference ships, 512 x 512 x 133, introduction.ipynb:230-272) with a 30-line NIfTI-1 reader
(gzip + the 348-byte header: nibabel / torchio are not installed here), keeps every second voxel
in x and y (nearest-neighbour down-sampling of a label map) and stores it as uint8:
tests/golden/reference_mask_ds2.npz.  tools/channels_bench.py repeats it 2 x 2 back to the
published size, so a ray meets label runs of the original lengths."""
import gzip
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/diffdrr/data/mask.nii.gz"

raw = gzip.open(SRC, "rb").read()
assert struct.unpack_from("<i", raw, 0)[0] == 348, "not a little-endian NIfTI-1 file"
dim = struct.unpack_from("<8h", raw, 40)
datatype, bitpix = struct.unpack_from("<hh", raw, 70)
vox_offset = int(struct.unpack_from("<f", raw, 108)[0])
pixdim = struct.unpack_from("<8f", raw, 76)
dtype = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 512: np.uint16}[datatype]
nx, ny, nz = dim[1:4]
vol = np.frombuffer(raw, dtype=dtype, count=nx * ny * nz, offset=vox_offset).reshape((nx, ny, nz), order="F")
labels = np.ascontiguousarray(vol[::2, ::2, :]).astype(np.uint8)
present = np.unique(vol)
print(f"{SRC}: {nx} x {ny} x {nz} {np.dtype(dtype).name}, spacing {pixdim[1:4]}, labels present "
      f"{len(present)} (max {present.max()}), background {float((vol == 0).mean()):.2%}")
out = os.path.join(HERE, "reference_mask_ds2.npz")
np.savez_compressed(out, labels=labels, spacing=np.asarray(pixdim[1:4], np.float32),
                    full_shape=np.asarray([nx, ny, nz]), n_labels=np.asarray(int(present.max()) + 1))
print(f"{out}: {labels.shape} uint8, {os.path.getsize(out) / 1024:.1f} kB")
