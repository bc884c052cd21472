"""The deal of the 21 ring tiles of k_band_factor_mx to its four MFMA waves (kMxGroupTile, hyperslam_amd/csrc/kernels_factor_mx.hpp).

Groups 0, 1 share SIMD 0, groups 2, 3 SIMD 1. In phase PH (block row it = PH mod 16) the pivot row X_(it-1) sits at ring position
p_i = 6 PH - 6 and its trailing band covers the offsets [12, 84): a tile (I, J) is updated iff both of its 16-position indices hold a
position of that range (15 of the 21 tiles in ten phases, all 21 in six). The search minimises, summed over the 16 phases, the tiles the
busier SIMD has to update (then the largest group, then the imbalance inside a SIMD).  usage: python tools/mx_tile_deal.py [seed]"""
import random
import sys

W = 96
tiles = [(I, J) for I in range(6) for J in range(I, 6)]


def active_indices(ph, hi=84):
    p_i = (6 * ph + W - 6) % W
    return {((p_i + off) % W) // 16 for off in range(12, hi)}


phases = []
for ph in range(16):
    a = active_indices(ph)
    phases.append([i for i, (I, J) in enumerate(tiles) if I in a and J in a])


def cost(assign):
    tot = mx = inner = 0
    for p in phases:
        cnt = [0] * 4
        for i in p:
            cnt[assign[i]] += 1
        s0, s1 = cnt[0] + cnt[1], cnt[2] + cnt[3]
        tot += max(s0, s1)
        mx = max(mx, s0, s1)
        inner += abs(cnt[0] - cnt[1]) + abs(cnt[2] - cnt[3])
    return (tot, mx, max(assign.count(k) for k in range(4)), inner)


random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
best = None
for _ in range(300000):
    a = [random.randrange(4) for _ in range(21)]
    c = cost(a)
    if best is None or c < best[0]:
        best = (c, a)
print("tiles inside the band per phase:", [len(p) for p in phases])
print("cost (sum over phases of the busier SIMD's tiles, max, largest group, imbalance):", best[0])
for k in range(4):
    print("group", k, [i for i in range(21) if best[1][i] == k], [tiles[i] for i in range(21) if best[1][i] == k])
