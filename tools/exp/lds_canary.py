"""Does an MLP kernel write into another workgroup's LDS?  Canary workgroups (one wave, a few KB of LDS with a pattern) run on one
stream while the deformation network's passes run on another (python tools/exp/lds_canary.py [fwd|bwd|both|none] [N])."""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import pkg  # noqa: E402

so = os.path.join(HERE, "lds_canary.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(HERE, "lds_canary.hip"), "-o", so])
lib = ctypes.CDLL(so)
what = sys.argv[1] if len(sys.argv) > 1 else "both"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
D = pkg("deform")
net = D.DeformModelNormal(is_blender=True, model_name="noise", device=torch.device("cuda:0"), trunk_impl="hip")
x = torch.randn(N, 3, device="cuda")
t = torch.tensor([[0.3]], device="cuda").expand(N, -1)
side = torch.cuda.Stream()
out = torch.zeros(8, dtype=torch.int32, device="cuda")
for words in (1152, 4096, 512):      # 4.6 KB like render_bwd3's, 16 KB, 2 KB
    for rep in range(6):
        out.zero_()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for j in range(4):
                if what == "none":
                    break
                if what == "fwd":
                    with torch.no_grad():
                        net.step_raw(x, t)
                else:
                    o = net.step_raw(x, t)
                    if what in ("bwd", "both"):
                        o.sum().backward()
        rc = lib.lds_canary_launch(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.c_void_p(out.data_ptr()), 256 * 6, 4000, words)
        assert rc == 0
        torch.cuda.synchronize()
        r = out.cpu().tolist()
        if r[0]:
            print(f"LDS words {words}, rep {rep}: {r[0]} corrupted reads in {r[1]} workgroups; e.g. word {r[2]} read {r[3]:#x} in block {r[4]}")
    print(f"LDS words {words}: done ({what})", flush=True)
