"""Debug / timing probe of the multi-phase propagation launch (spmm_lds_multi_kernel): one fused propagate on a golden graph
against one launch per layer, the sync words afterwards, and (--time) event-timed forward passes both ways.
    timeout 120 python scripts/lds_multi_probe.py [golden name] [layers] [--time]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import _tune  # noqa: E402,F401  (binds RECAD_TUNING_LIB's variant build, if set, before the product library is loaded)
from recad_amd import _lib, model  # noqa: E402
from tests import _golden as G  # noqa: E402
from tests._stub import LGN_KEYS, ReplayDataset  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    name = args[0] if args else "lightgcn_dev_d64"
    L = int(args[1]) if len(args) > 1 else 2
    dev = torch.device("cuda:0")
    if "--watch" in sys.argv:
        _lib.RK_LDS_SYNC_WORDS = 2560 + 4 * 256 + 64   # room for the debug build's per-workgroup markers
    g = G.load(name)
    outs = {}
    for fuse in (False, True):
        ds = ReplayDataset(g, LGN_KEYS, device=dev, steps=[0])
        m = model.from_config("victim", "lightgcn", latent_dim_rec=int(g["dim"]), lightGCN_n_layers=L).I(dataset=ds)
        m.use_lds, m.fuse_layers = True, fuse
        u0, i0 = G.lightgcn_init(g)
        m.embedding_user.weight.data.copy_(torch.from_numpy(u0))
        m.embedding_item.weight.data.copy_(torch.from_numpy(i0))
        m = m.to(dev)
        t0 = time.perf_counter()
        if fuse and "--watch" in sys.argv:
            # debug build (-DRK_LDS_DEBUG through RECAD_TUNING_LIB): launch on a side stream, watch it from the host, dump the markers if it hangs
            m._ensure_handle()
            big = m._ws["lds_sync"]
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                lu, li = m._propagate()
            for k in range(40):
                time.sleep(0.5)
                if side.query():
                    print("  completed in < %.1f s" % (0.5 * (k + 1)), flush=True)
                    break
            else:
                copy = torch.cuda.Stream()
                host = torch.empty(big.numel(), dtype=torch.int32).pin_memory()
                with torch.cuda.stream(copy):
                    host.copy_(big, non_blocking=True)
                for k in range(20):
                    time.sleep(0.5)
                    if copy.query():
                        break
                s = host.numpy()
                print("HUNG after 20 s.  heads", s[0:256:32].tolist(), "arrive", s[256:256 + 16 * 32:32].tolist(), "done", int(s[2304]), "err", int(s[2336]), flush=True)
                mk = s[2560:2560 + 4 * 256].reshape(256, 4)
                import collections
                print("stage histogram", dict(collections.Counter(mk[:, 0].tolist())), flush=True)
                for b in range(0, 256, 8):
                    print("  wg", b, mk[b].tolist(), flush=True)
                import os
                os._exit(3)
        lu, li = m.computer()
        torch.cuda.synchronize()
        print(f"fuse={fuse}: first propagate {time.perf_counter() - t0:.3f} s", flush=True)
        outs[fuse] = torch.cat([lu, li]).cpu().numpy()
        sync = m._ws.get("lds_sync")
        if sync is not None:
            s = sync.cpu().numpy()
            print("  heads", s[0:256:32].tolist(), "arrive", s[256:256 + 16 * 32:32].tolist(), "done", int(s[2304]), "err", int(s[2336]), flush=True)
            plan = m._ws["lds"][0].cpu().numpy()
            o = int(plan[17])
            print("  mq header", plan[o:o + 4].tolist(), "queues", plan[o + 4:o + 4 + 2 * int(plan[o])].tolist(), "members",
                  plan[o + 4 + 2 * int(plan[o]):o + 4 + 2 * int(plan[o]) + int(plan[o + 1])].tolist(), flush=True)
        if "--time" in sys.argv:
            h = m._ensure_handle()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(5):
                _lib.check(_lib.lib().rk_lightgcn_propagate(h, _lib.stream_ptr()), "propagate")
            e0.record()
            for _ in range(100):
                _lib.check(_lib.lib().rk_lightgcn_propagate(h, _lib.stream_ptr()), "propagate")
            e1.record()
            torch.cuda.synchronize()
            print(f"  propagate (pack + {L} layers): {e0.elapsed_time(e1) * 10:.2f} us", flush=True)
    print("identical bits:", bool(np.array_equal(outs[False], outs[True])), "max abs diff", float(np.abs(outs[False] - outs[True]).max()))


if __name__ == "__main__":
    main()
