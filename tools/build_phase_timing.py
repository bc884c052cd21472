"""Phase timestamps of k_build_visual (HS_DEBUG_FLAGS=32, profiling build: tools/build_profiling_lib.sh).
usage (GPU box): python tools/build_phase_timing.py [config=1|2|3|r]"""
import os
import sys, ctypes as C; sys.path.insert(0, ".")
import numpy as np
os.environ["HS_DEBUG_FLAGS"] = str(32 | int(os.environ.get("HS_DEBUG_FLAGS", "0")))
import hyperslam_amd as ha
from hyperslam_amd import synthetic, _lib
_lib.PRODUCT_LIB = os.path.join("tools", "libhyperslam_hip_prof.so")
# config "r": the replay's steady state — ~34 control points, 480 landmarks tracked over up to the whole window (bw ~ 33: window-wide band)
cfg = sys.argv[1] if len(sys.argv) > 1 else "1"
w = synthetic.small_visual(order=4, n_cp=36, n_landmarks=480, obs_pairs=18, span=3.0) if cfg == "r" else \
    {1: synthetic.config1, 2: synthetic.config2, 3: synthetic.config3}[int(cfg)]()
p = ha.Problem(w); p.snapshot()
for i in range(3): p.restore(); s = p.solve(1)
lib = _lib.load().cdll
n = 48 * 1024 + 64 * 1024
buf = np.zeros(n, np.int64)
lib.hs_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.hs_debug_read(p.h, buf.ctypes.data, n)
t = buf[48 * 1024:].reshape(1024, 4, 16)
ok = t[:, 0, 0] > 0
t = t[ok]
print("chunks stamped:", len(t), " residuals / landmarks per chunk (median, max):", np.median(t[:, 0, 13]), t[:, 0, 13].max(), np.median(t[:, 0, 14] & 0xffff), (t[:, 0, 14] & 0xffff).max())
names = ["start", "inputs + keys (b)", "sorted, pos (b)", "linearised", "barrier", "H/b, W blocks", "J'J tiles done", "barrier", "Cholesky (b)", "Y-hat (b)",
         "Yh Yh' done", "written", "cost summed"]  # (b): stamp taken behind the phase's barrier
base = t[:, :, 0].min(axis=1)[:, None, None]
rel = (t[:, :, :13] - base) * 0.01  # 100 MHz clock -> us
print("median over chunks [us after the chunk's first wave started]; columns = wave 0..3")
for i, nme in enumerate(names):
    print(f"{i:2d} {nme:16s}", " ".join(f"{np.median(rel[:, wv, i]):7.2f}" for wv in range(4)), "   max over chunks/waves", f"{rel[:, :, i].max():7.2f}")
t0 = t[:, :, 0].min()
print("kernel span over the stamped chunks [us]: first start -> last end", (t[:, :, 12].max() - t0) * 0.01, " chunk starts (min, median, max)",
      (np.min(t[:, 0, 0]) - t0) * 0.01, (np.median(t[:, 0, 0]) - t0) * 0.01, (np.max(t[:, 0, 0]) - t0) * 0.01)

end = (t[:, :, 12].max(axis=1) - t0) * 0.01
start = (t[:, :, 0].min(axis=1) - t0) * 0.01
print("all chunks: start quantiles [us]", np.percentile(start, [0, 25, 50, 75, 100]).round(2), " end quantiles", np.percentile(end, [0, 25, 50, 75, 100]).round(2),
      " duration quantiles", np.percentile(end - start, [0, 25, 50, 75, 100]).round(2))
hw = t[:, 0, 15] & 0xffffffff
xcc = (t[:, 0, 15] >> 32) & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 0x1; se = (hw >> 13) & 0x7
key = xcc * 10000 + se * 1000 + sh * 100 + cu
uniq, cnt = np.unique(key, return_counts=True)
print("distinct (xcc, se, sh, cu):", len(uniq), " chunks per CU: max", cnt.max(), " histogram", np.bincount(cnt))
# overlap: for every CU the number of chunks alive at the median start of its second chunk
over = 0
for k_ in uniq:
    idx = np.where(key == k_)[0]
    if len(idx) >= 2:
        o = np.argsort(start[idx])
        a, b = idx[o[0]], idx[o[1]]
        over += int(start[b] < end[a] - 1.0)
print("CUs whose second chunk started more than 1 us before the first ended (co-resident workgroups):", over, "of", int((cnt >= 2).sum()))
# which SIMD each wave of a workgroup runs on, and whether the two co-resident workgroups of a CU put wave k on the same SIMD
simd = (t[:, :, 15] & 0xffffffff) >> 4 & 3
print("SIMD of waves 0..3, first 6 chunks:", simd[:6].tolist())
same = tot = 0
pat = {}
for k_ in uniq:
    idx = np.where(key == k_)[0]
    if len(idx) == 2:
        tot += 1
        same += int((simd[idx[0]] == simd[idx[1]]).all())
        pat[tuple(simd[idx[0]].tolist() + simd[idx[1]].tolist())] = pat.get(tuple(simd[idx[0]].tolist() + simd[idx[1]].tolist()), 0) + 1
print("CUs with two chunks:", tot, " wave k of both on the same SIMD:", same, " patterns (wg A waves | wg B waves):", sorted(pat.items(), key=lambda kv: -kv[1])[:6])

# the slowest chunks: first control point, records, landmarks, phase lengths (wave-max) — which phase makes the tail
dur = (t[:, :, 12].max(axis=1) - t[:, :, 0].min(axis=1)) * 0.01
ph = (t[:, :, 1:13].max(axis=1) - t[:, :, 0:12].max(axis=1)) * 0.01
order = np.argsort(-dur)
print("slowest chunks: duration | cf records landmarks | phase lengths (stamps 1..12)")
for i in list(order[:8]) + list(order[len(order) // 2: len(order) // 2 + 3]):
    print(f"  {dur[i]:6.2f} | cf {int(t[i, 0, 14]) >> 16:4d} n {int(t[i, 0, 13]):4d} l {int(t[i, 0, 14]) & 0xffff:3d} |", " ".join(f"{x:5.2f}" for x in ph[i]))
cfs = t[:, 0, 14] >> 16
for lo, hi in ((0, 2), (2, 10), (10, 100), (100, 120), (120, 200)):
    m = (cfs >= lo) & (cfs < hi)
    if m.any(): print(f"chunks with cf in [{lo}, {hi}): {int(m.sum()):4d}  median duration {np.median(dur[m]):6.2f}  max {dur[m].max():6.2f}")
# which workgroup indices sit alone on their CU (dispatch order -> placement): the slots for the heaviest chunks
lone = sorted(int(np.where(key == k_)[0][0]) for k_ in uniq[cnt == 1])
print("workgroups alone on a CU:", len(lone), " indices", lone[:5], "...", lone[-5:], " contiguous:", lone == list(range(lone[0], lone[0] + len(lone))) if lone else None)
pairs = [tuple(sorted(np.where(key == k_)[0].tolist())) for k_ in uniq[cnt == 2]]
d = np.array([b - a for a, b in pairs])
if len(d): print("index distance of the two workgroups of a CU: min", d.min(), " median", np.median(d), " max", d.max(), " share == 256:", float((d == 256).mean()))
# where the slow chunks run: per XCC / per shader engine, and whether the two workgroups of a CU are slow together
slow = dur > np.percentile(dur, 50) + 2.5
print(f"slow chunks (> median + 2.5 us): {int(slow.sum())} of {len(dur)}")
print("  per XCC   (slow / all):", " ".join(f"{int(slow[xcc == x].sum())}/{int((xcc == x).sum())}" for x in range(8)))
print("  per SE    (slow / all):", " ".join(f"{int(slow[se == x].sum())}/{int((se == x).sum())}" for x in range(8)))
print("  per CU id (slow / all):", " ".join(f"{int(slow[cu == x].sum())}/{int((cu == x).sum())}" for x in range(16)))
both = one = 0
for k_ in uniq:
    idx = np.where(key == k_)[0]
    if len(idx) == 2: both += int(slow[idx].all()); one += int(slow[idx].any() and not slow[idx].all())
print("  CUs with two chunks: both slow", both, " exactly one slow", one, "  slow chunks alone on their CU:", int(sum(slow[np.where(key == k_)[0]].any() for k_ in uniq[cnt == 1])))
wgid = np.where(ok)[0]
print("  workgroup index of the slow chunks (mod 256) histogram by 32:", np.bincount((wgid[slow] % 256) // 32, minlength=8))
print("  median duration by records-in-K-segments weight is not available here; phase medians of slow vs other chunks:")
print("   slow :", " ".join(f"{x:5.2f}" for x in np.median(ph[slow], axis=0)))
print("   other:", " ".join(f"{x:5.2f}" for x in np.median(ph[~slow], axis=0)))
print("  slow chunks with workgroup index < 256:", int((wgid[slow] < 256).sum()), " >= 256:", int((wgid[slow] >= 256).sum()))
part = {}
for k_ in uniq:
    idx = np.where(key == k_)[0]
    if len(idx) == 2: part[idx[0]] = idx[1]; part[idx[1]] = idx[0]
sl = [i for i in np.where(slow)[0] if i in part]
print("  slow chunk vs its CU partner, medians: duration %.2f / %.2f  start %.2f / %.2f  records %d / %d" % (np.median(dur[sl]), np.median([dur[part[i]] for i in sl]),
      np.median(start[sl]), np.median([start[part[i]] for i in sl]), np.median(t[sl, 0, 13]), np.median([t[part[i], 0, 13] for i in sl])))
print("  stamps of the slow chunk minus its partner's, median per stamp (us):", " ".join(f"{x:5.2f}" for x in np.median([(t[i, :, :13].max(axis=0) - t[part[i], :, :13].max(axis=0)) * 0.01 for i in sl], axis=0)))
nsl = [i for i in np.where(~slow)[0] if i in part and not slow[part[i]]]
print("  pairs without a slow chunk:", len(nsl) // 2, " workgroup indices (mod 256) histogram by 32:", np.bincount((wgid[nsl] % 256) // 32, minlength=8) // 2)
