"""Which ReLU gates of the NCF tower (factor 256, L = 3, the reference's golden batch) land on the other side of zero than fp64
puts them, as a function of the fp32 summation order of the forward: one k-ordered chain (what the MFMA does over a whole K),
8 k-blocks combined pairwise (csrc/ncf.hip gemm_fwd_blocked), blocks of kc added in sequence, 16 interleaved lanes + tree, fp64
accumulation -- each feeding ITS OWN activations to the next layer -- and ATen (torch CPU, the reference's arithmetic).
CPU only; builds a small C helper with gcc.  Round-5 result: chain [0, 1, 0] flips (|z64| = 6.8e-10, rms 9.4e-3),
8 blocks / fp64 accumulation / ATen [0, 0, 0]."""
import ctypes as C, os, subprocess, sys, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import _golden as G  # noqa: E402

SRC = r'''
#include <stdint.h>
#include <stddef.h>
#include <math.h>
void layer(int B, int in, int out, const float *x, const float *w, float *z, int mode, int kc)
{
    for (int b = 0; b < B; ++b)
        for (int r = 0; r < out; ++r) {
            const float *xv = x + (size_t)b * in, *wv = w + (size_t)r * in;
            float s = 0.f;
            if (mode == 0) { for (int k = 0; k < in; ++k) s = fmaf(wv[k], xv[k], s); }
            else if (mode == 1) {
                float p[8]; int blk = in / 8;
                for (int q = 0; q < 8; ++q) { float t = 0.f; for (int k = q * blk; k < (q + 1) * blk; ++k) t = fmaf(wv[k], xv[k], t); p[q] = t; }
                s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
            } else if (mode == 2) {
                for (int k0 = 0; k0 < in; k0 += kc) { float t = 0.f; for (int k = k0; k < k0 + kc && k < in; ++k) t = fmaf(wv[k], xv[k], t); s += t; }
            } else if (mode == 4) {
                float p[16]; for (int j = 0; j < 16; ++j) p[j] = 0.f;
                for (int k = 0; k < in; ++k) p[k & 15] = fmaf(wv[k], xv[k], p[k & 15]);
                for (int st = 8; st > 0; st >>= 1) for (int j = 0; j < st; ++j) p[j] += p[j + st];
                s = p[0];
            } else { double t = 0.0; for (int k = 0; k < in; ++k) t += (double)wv[k] * (double)xv[k]; s = (float)t; }
            z[(size_t)b * out + r] = s;
        }
}
'''
d = tempfile.mkdtemp()
open(os.path.join(d, "g.c"), "w").write(SRC)
subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", os.path.join(d, "g.c"), "-o", os.path.join(d, "g.so"), "-lm"])
lib = C.CDLL(os.path.join(d, "g.so"))
g = G.load(sys.argv[1] if len(sys.argv) > 1 else "ncf_game_f256_l3")
f, L = int(g["factor"]), int(g["layers"])
(ug, ig, um, im), W, b, pw, pb = G.ncf_init(g)
n0 = int(g["batch_len"][0])
u, i = (g["batches"][0, k, :n0].astype(np.int64) for k in range(2))
x0 = np.concatenate([um[u], im[i]], 1).astype(np.float32)
x64, z64s = x0.astype(np.float64), []
for l in range(L):
    z = x64 @ W[l].astype(np.float64).T + b[l]
    z64s.append(z)
    x64 = np.maximum(z, 0)
for l in range(L):
    a = np.abs(z64s[l])
    print("layer %d: rms %.3e, smallest |z64| %.3e" % (l, np.sqrt((z64s[l] ** 2).mean()), a.min()))


def fwd(mode, kc=0):
    x, zs = x0, []
    for l in range(L):
        inn, out = W[l].shape[1], W[l].shape[0]
        z = np.empty((n0, out), dtype=np.float32)
        lib.layer(n0, inn, out, np.ascontiguousarray(x).ctypes.data_as(C.c_void_p), np.ascontiguousarray(W[l]).ctypes.data_as(C.c_void_p),
                  z.ctypes.data_as(C.c_void_p), mode, kc)
        z = z + b[l]
        zs.append(z)
        x = np.maximum(z, 0)
    return zs


for label, mode, kc in (("one chain", 0, 0), ("8 blocks pairwise", 1, 0), ("kc = 256 in sequence", 2, 256), ("16 lanes + tree", 4, 0), ("fp64 accumulation", 3, 0)):
    zs = fwd(mode, kc)
    print("%-22s gates != fp64 per layer %s  max |z - z64| / rms %s" % (label, [int(((a > 0) != (c > 0)).sum()) for a, c in zip(zs, z64s)],
                                                                      ["%.1e" % (np.abs(a - c).max() / np.sqrt((c ** 2).mean())) for a, c in zip(zs, z64s)]))
torch.set_num_threads(1)
x, fl = torch.from_numpy(x0), []
for l in range(L):
    z = torch.nn.functional.linear(x, torch.from_numpy(W[l]), torch.from_numpy(b[l]))
    fl.append(int(((z.numpy() > 0) != (z64s[l] > 0)).sum()))
    x = torch.relu(z)
print("%-22s gates != fp64 per layer %s" % ("ATen (torch CPU)", fl))
