"""LDS bank model of conv3d_k3_wino2d_kernel's V-tile accesses (sceneego_amd/csrc/conv3d_wino2d.hip), per MI355X_MICROARCH.md "LDS":

  ds_read_b128   4 passes of 16 lanes {0-3,12-15,20-27} {4-11,16-19,28-31} {32-35,44-47,52-59} {36-43,48-51,60-63}, bank = dword % 64
  ds_write_b128  8 passes of 8 consecutive lanes, bank = dword % 32

For the layout constants given on the command line (default: the kernel's) it prints the extra LDS cycles per step (one 8-channel
chunk of one tile, both wave groups) of (1) the MFMA phase's B-operand reads and (2) the 96 stores of the V-tile transform, with the
lanes that have no output parked as the kernel parks them.  Round 2 (VREC 200, no tile pad, dummy = lane * 4 floats): 792 extra
cycles per step, next to the 804 the counters showed.  CAUTION: the model explains WHERE the conflicts are (the counter attribution
of tools/pmc_lds_attr.sh agrees: all of them come from these stores) but not their number after a layout change - for the round-3
layout it predicts 96, the counter says 590-710 (profiles/r03_lds_conflict_attribution.txt): ds_write_b128 does not bank the way
the published table says.

usage: python tools/lds_conflicts_w2d.py [VREC TILE_PAD [old-dummy]]
"""
import re
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READ_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
READ_GROUPS += [[l + 32 for l in g] for g in READ_GROUPS]
CHUNK = 18432


def kernel_constants():
    src = open(os.path.join(ROOT, "sceneego_amd", "csrc", "conv3d_wino2d.hip")).read()
    vrec = int(re.search(r"constexpr int W2_VREC = (\d+);", src).group(1))
    m = re.search(r"constexpr int W2_VTILE = 18 \* W2_VREC(?: \+ (\d+))?;", src)
    return vrec, int(m.group(1) or 0)


def extra_cycles(addrs, groups, banks):
    """addrs: 64 dword addresses of a 16-byte access; extra LDS cycles = sum over passes of (max distinct addresses on a bank - 1)."""
    extra = 0
    for g in groups:
        per = {}
        for l in g:
            for d in range(4):
                per.setdefault((addrs[l] + d) % banks, set()).add(addrs[l] + d)
        extra += max(len(v) for v in per.values()) - 1
    return extra


def model(vrec, pad, old_dummy=False):
    vtile = 18 * vrec + pad
    vg = 2 * vtile
    dummy_base = CHUNK + 2 * vg
    reads = stores = 0
    write_groups = [list(range(8 * k, 8 * k + 8)) for k in range(8)]
    for G in range(2):
        for wq in range(4):
            jt = wq & 1
            # B-operand reads: b_dx[dx] + (xz * 2 + q) * 16, lane (px, h)
            for dx in range(3):
                base = [CHUNK + G * vg + jt * vtile + ((l & 15) + dx) * vrec + (l >> 4) * 4 for l in range(64)]
                for xz in range(6):
                    for q in range(2):
                        reads += extra_cycles([b + (xz * 2 + q) * 16 for b in base], READ_GROUPS, 64)
            # V-tile transform stores
            for which in ("pq", "rs"):
                off, has = [], []
                for lane in range(64):
                    i16 = lane & 15
                    sk = i16 % 3
                    stask = (wq * 4 + (lane >> 4)) * 5 + i16 // 3
                    s_on = i16 < 15 and stask < 72
                    sxx, sp = (stask >> 2 if s_on else 0), stask & 3
                    if which == "pq":
                        has.append(s_on and sk < 2)
                        off.append(CHUNK + G * vg + sk * vtile + sxx * vrec + sp * 4)
                    else:
                        has.append(s_on and sk > 0)
                        off.append(CHUNK + G * vg + (sk - 1) * vtile + sxx * vrec + sp * 4 + 16)
                addr = []
                for lane in range(64):
                    if has[lane]:
                        addr.append(off[lane])
                    elif old_dummy:
                        addr.append(dummy_base + lane * 4)
                    else:       # the kernel's dummy_slot(): rank-th free 16-byte slot of the lane's 8-lane pass
                        grp = range(lane & ~7, (lane & ~7) + 8)
                        used = {(off[j] >> 2) & 7 for j in grp if has[j]}
                        rank = sum(1 for j in grp if not has[j] and j < lane)
                        free = [k for k in range(8) if k not in used]
                        addr.append(dummy_base + (free[min(rank, len(free) - 1)] if free else lane & 7) * 4)
                for z in range(6):
                    stores += extra_cycles([a + z * 32 for a in addr], write_groups, 32)
    return reads, stores


if __name__ == "__main__":
    if len(sys.argv) > 2:
        vrec, pad = int(sys.argv[1]), int(sys.argv[2])
    else:
        vrec, pad = kernel_constants()
    r, s = model(vrec, pad, old_dummy=len(sys.argv) > 3)
    print(f"VREC {vrec}, tile pad {pad}: extra LDS cycles per step - operand reads {r} (of {2 * 4 * 36 * 4} conflict-free), V-tile stores {s} "
          f"(of {96 * 8} conflict-free)")
