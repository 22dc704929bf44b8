"""LDS bank-conflict census for the conv_gemm_v3 activation patch (design aid, host only).

Patch layout: 32-byte physical rows (16 channels), frame p at p*S rows, a frame's rows split into `FM` parity planes
(row r -> plane r % FM at offset PLo[plane], index r // FM).  A ds_read_b128 of the MFMA B operand (16x16x32 bf16) has lane
l = (col c = l & 15, k group g = l >> 4): g >> 1 selects the tap of the pair, g & 1 the 8-channel piece.
ds_read_b128 is served in four 16-lane groups (MI355X_MICROARCH.md, LDS); bank = (byte address / 4) % 64.
"""
import itertools, sys

GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
          [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
          [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59],
          [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63]]


def cycles(addrs):
    """LDS cycles of one ds_read_b128 wave instruction given 64 byte addresses (4 = conflict free)."""
    tot = 0
    for g in GROUPS:
        cnt = {}
        for l in g:
            a = addrs[l]
            for k in range(4):
                b = ((a >> 2) + k) & 63
                cnt.setdefault(b, set()).add(a >> 2 << 2 | 0 if False else (a + 4 * k))
        tot += max(len(v) for v in cnt.values())
    return tot


def frame_of(J, mi, q):
    """frame (relative to the wave's first) of column quad/half q of 16-column tile mi"""
    if J >= 16:
        return mi // (J // 16)
    if J == 8:
        return (mi >> 1) * 4 + (mi & 1) + 2 * q
    if J == 4:
        return (mi >> 1) * 8 + (mi & 1) * 2 + (0, 1, 5, 4)[q]
    raise ValueError


def census(J, FM, NF, S, planes):
    FR = (J - 1) * FM + NF
    worst, tot, n = 0, 0, 0
    for mi in range(8):
        for j in range(NF):
            its = (2 * j, 2 * j + 1)
            addrs = []
            for l in range(64):
                c, g = l & 15, l >> 4
                it = its[g >> 1]
                kt, tap = divmod(it, NF)
                if J >= 16:
                    fr = frame_of(J, mi, 0); jl = (mi % (J // 16)) * 16 + c
                else:
                    fr = frame_of(J, mi, c // J); jl = c % J
                r = jl * FM + tap
                prow = (fr + kt) * S + planes[r % FM] + r // FM
                addrs.append(prow * 32 + (g & 1) * 16)
            cy = cycles(addrs)
            worst = max(worst, cy); tot += cy; n += 1
    return tot / n, worst


if __name__ == "__main__":
    for (FM, NF) in ((2, 5), (1, 3), (1, 2)):
        for J in (4, 8, 16, 32):
            FR = (J - 1) * FM + NF
            best = None
            for S in range(FR, FR + 9):
                for pad in range(0, 3):
                    planes = [0, (FR + 1) // 2 + pad] if FM == 2 else [0]
                    if FM == 2 and planes[1] + FR // 2 > S:
                        continue
                    avg, worst = census(J, FM, NF, S, planes)
                    if best is None or avg < best[0] - 1e-9:
                        best = (avg, worst, S, planes)
            TB = 256 // J
            print(f"FM={FM} NF={NF} J={J:2d} FR={FR:2d}: best avg {best[0]:.2f} worst {best[1]} at S={best[2]} planes={best[3]}"
                  f"  patch half {(TB + 1) * best[2] * 32 / 1024:.1f} KB; unpadded S={FR}: {census(J, FM, NF, FR, [0, (FR + 1) // 2] if FM == 2 else [0])}")
