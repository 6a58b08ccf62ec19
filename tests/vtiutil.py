"""Reader of the reference's hand-written image-data files (.vti, appended raw doubles): {name: array[z, y, x]} + the extent."""
import re

import numpy as np


def read_vti(path):
    blob = open(path, "rb").read()
    head_end = blob.index(b"<AppendedData encoding=\"raw\">")
    head = blob[:head_end].decode()
    ext = [int(x) for x in re.search(r'<Piece Extent="([-0-9 ]+)"', head).group(1).split()]
    nx, ny, nz = ext[1] - ext[0] + 1, ext[3] - ext[2] + 1, ext[5] - ext[4] + 1
    start = blob.index(b"_", head_end) + 1
    out = {}
    for m in re.finditer(r'<DataArray type="Float64" Name="(\w+)" format="appended" offset="(\d+)"', head):
        off = start + int(m.group(2))
        nbytes = int(np.frombuffer(blob, dtype="<u4", count=1, offset=off)[0])
        assert nbytes == 8 * nx * ny * nz, (path, m.group(1), nbytes, nx, ny, nz)
        out[m.group(1)] = np.frombuffer(blob, dtype="<f8", count=nx * ny * nz, offset=off + 4).reshape(nz, ny, nx)
    return out, ext
