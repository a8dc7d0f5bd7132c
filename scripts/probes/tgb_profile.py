"""Where a tg_bwd_kernel workgroup spends its cycles (csrc/train_bwd.hip).  Needs a tuning build of the library:
    PN2_EXTRA_HIPCC_FLAGS=-DPN2_TGB_PROFILE python -c "from hotrack_amd import _build; _build.build(force=True)"
    python scripts/probes/tgb_profile.py
Prints, per layer shape, the mean cycle counts over all workgroups of wave 0 and, per wave, of the two matrix phases and the two
barrier waits: prologue | commit | barrier 1 | data gradient | weight gradient | barrier 2 | epilogue | tail.  Rebuild without the
flag afterwards."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "network")]
import torch  # noqa: E402

SHAPES = [("sa1 32->64", 32 * 256 * 32, 32, 64), ("sa2 64->128", 32 * 128 * 32, 64, 128), ("q64 128->192", 32 * 21 * 64, 128, 192),
          ("q16 128->192", 32 * 21 * 16, 128, 192), ("q64 128->128", 32 * 21 * 64, 128, 128)]


def main():
    from hotrack_amd import train_stack
    from hotrack_amd.train_ops import Workspace
    lib = train_stack._lib
    if not hasattr(lib, "pn2x_tg_bwd_set_profile"):
        raise SystemExit("library built without -DPN2_TGB_PROFILE")
    lib.pn2x_tg_bwd_set_profile.argtypes = [ctypes.c_void_p]
    lib.pn2x_tg_bwd_set_profile.restype = None
    prof = torch.zeros(1024 * 8 * 8 + 1024 * 8 * 4, dtype=torch.int64, device="cuda")  # + the prologue sub-phases of the round-5 kernel
    lib.pn2x_tg_bwd_set_profile(prof.data_ptr())
    names = ["prologue", "commit", "barrier1", "dgrad", "wgrad", "barrier2", "epilogue", "tail"]
    out = {}
    for name, R, cin, cout in SHAPES:
        K = 16
        convs = [torch.nn.Conv1d(cin, cout, 1).cuda()]
        bns = [torch.nn.BatchNorm1d(cin).cuda().train(), torch.nn.BatchNorm1d(cout).cuda().train()]
        ws = Workspace("cuda")
        y = torch.randn(R, cin, device="cuda").requires_grad_(True)
        layers = [train_stack.Layer(None, bns[0]), train_stack.Layer(convs[0].weight, bns[1], convs[0].bias)]
        o = train_stack.mlp_stack(y, layers, ws, max_over=K)
        for _ in range(3):
            prof.zero_()
            ws_gen = torch.autograd.grad(o, y, torch.ones_like(o), retain_graph=True)
        torch.cuda.synchronize()
        grid = int(lib.pn2x_tg_bwd_partials(R, cout, cin))
        pw = prof[:1024 * 64].view(-1, 8, 8)[:min(grid, 1024)].double()   # [workgroup][wave][phase]
        pw = pw[pw[:, 0].sum(dim=1) > 0]
        p = pw[:, 0]
        p2 = prof[1024 * 64:].view(-1, 8, 4)[:min(grid, 1024)].double()
        pro = {n: [round(float(p2[:, w, i].mean())) for w in (0, 3, 7)] for i, n in enumerate(("loads_issued", "constants", "w_staged", "barrier"))}
        out[name] = {"prologue_parts_waves_0_3_7": pro, "workgroups": int(p.shape[0]), "tiles": (R + 63) // 64,
                     **{n: round(float(p[:, i].mean())) for i, n in enumerate(names)}, "total": round(float(p.sum(dim=1).mean())),
                     "per_wave": {n: [round(float(pw[:, w, i].mean())) for w in range(8)] for i, n in enumerate(names) if n in ("barrier1", "dgrad", "wgrad", "barrier2", "commit", "epilogue")}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
