"""debug: fused 2-D lattice launch against the two-launch path (WX_KNOBS=1 WX_L2D_FUSED=0 in a child process)"""
import os, sys, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def run(m, L, B, inverse):
    import torch
    import waveletsext_jl_amd as wx
    wt = wx.wavelet(wx.WT.db4)
    x = wx.jl_empty((m, m, B), torch.float32, "cuda")
    x.normal_(generator=torch.Generator(device="cuda").manual_seed(B))
    outs = []
    for rep in range(3):
        y = wx.iwptall(x, wt, L) if inverse else wx.wptall(x, wt, L)
        outs.append(y.clone())
        # dirty the scratch pool between the calls
        junk = torch.full((m * m * B + 4096,), 7.0, device="cuda")
        del junk
    return [o.cpu().numpy() for o in outs]


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        m, L, B, inv = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
        outs = run(m, L, B, inv)
        np.save(sys.argv[6], np.stack(outs))
        sys.exit(0)
    for (m, L, B) in ((256, 5, 2), (256, 5, 5), (256, 5, 64), (512, 6, 3), (512, 6, 200), (512, 6, 1024), (1024, 7, 2), (1024, 7, 70)):
        for inv in (0, 1):
            res = {}
            for fused in (1, 0):
                env = dict(os.environ, WX_KNOBS="1", WX_L2D_FUSED=str(fused))
                f = "/tmp/fd_%d.npy" % fused
                subprocess.check_call([sys.executable, __file__, "child", str(m), str(L), str(B), str(inv), f], env=env)
                res[fused] = np.load(f)
            ref = res[0][0]
            line = "m=%d B=%d inv=%d:" % (m, B, inv)
            for rep in range(3):
                d = np.abs(res[1][rep] - ref)
                bad_img = np.nonzero(d.max(axis=(0, 1)) > 1e-5 * np.abs(ref).max())[0]
                line += "  rep%d bad images %d/%d" % (rep, len(bad_img), B)
                if len(bad_img) and rep == 0:
                    b = bad_img[0]
                    rows = np.nonzero(d[:, :, b].max(axis=1) > 1e-5)[0]
                    cols = np.nonzero(d[:, :, b].max(axis=0) > 1e-5)[0]
                    line += " [first bad img %d: rows %d..%d (%d), cols %d..%d (%d)]" % (b, rows.min(), rows.max(), len(rows), cols.min(), cols.max(), len(cols))
            line += "  unfused deterministic %s" % bool((res[0][0] == res[0][2]).all())
            print(line, flush=True)
