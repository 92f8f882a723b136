"""Histogram of workgroups per CU for partially filled grids (see dispatch_probe.hip)."""
import collections
import ctypes
import os
import subprocess
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "dispatch_probe.so")
if not os.path.exists(so):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so,
                    os.path.join(here, "dispatch_probe.hip")], check=True)
lib = ctypes.CDLL(so)
for blocks, lds in ((432, 40960), (432, 73728), (576, 40960), (1728, 40960), (256, 40960)):
    out = torch.zeros(blocks * 6, dtype=torch.int32, device="cuda")
    rc = lib.run_probe(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.c_void_p(out.data_ptr()),
                       blocks, lds, 200000)
    torch.cuda.synchronize()
    o = out.cpu().view(blocks, 6).numpy().astype("uint32")
    per_cu = collections.Counter()
    for b in range(blocks):
        xcc, hw = int(o[b, 0]) & 0xF, int(o[b, 1])
        cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
        per_cu[(xcc, se, sh, cu)] += 1
    hist = collections.Counter(per_cu.values())
    t0 = (o[:, 3].astype("uint64") << 32) | o[:, 2]
    print("blocks=%d lds=%d rc=%d: distinct CUs used=%d, blocks-per-CU histogram=%s, xcc(b) == b%%8 for %d/%d blocks, "
          "start spread=%d cycles, block duration ~%d cycles" % (
              blocks, lds, rc, len(per_cu), dict(sorted(hist.items())),
              sum(1 for b in range(blocks) if (int(o[b, 0]) & 0xF) == b % 8), blocks,
              int(t0.max() - t0.min()), int(o[:, 4].mean())))
