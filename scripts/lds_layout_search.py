#!/usr/bin/env python3
"""Offline search of the LDS patch layouts used by csrc/resnet_ws.h (ds_read_b32: a wave is served in two 32-lane halves,
bank = dword address mod 32; lanes l and l+32 never conflict).  Per conv geometry (input size HIN, stride S) it picks the band
size BPOS (output positions per workgroup pass), the M-tile shape (16 positions = TR rows x TC columns of the output map, or
several whole images for the tiny maps), the patch row stride RS, the per-image region stride ISZ and the channel-plane stride
PS such that the A-operand gather of every M-tile is bank-conflict free and the patch fits 80 KiB (two workgroups per CU).
The same predicate is re-checked at compile time by static_asserts in the header."""


def geo(HIN, S, BPOS, TC, kind=0):
    """kind 0: 3x3 pad-1 convolution input patch; kind 1: patch of the dy map for the stride-2 data gradient (a 2x2-tap
    stride-1 gather with a halo row / column at the bottom / right only; HIN = size of the dy map, S must be 1)."""
    HO = HIN // S
    WO, PI = HO, HO * HO
    multi = PI < BPOS
    NI = BPOS // PI if multi else 1
    RB = HO if multi else BPOS // WO
    PR = S * (RB - 1) + 3 if kind == 0 else RB + 1
    return dict(HIN=HIN, S=S, BPOS=BPOS, HO=HO, WO=WO, PI=PI, multi=multi, NI=NI, RB=RB, PR=PR, TC=TC, kind=kind)


def tile_pos(g, t, lr):
    """(image-in-band, oy-in-band, ox) of lane row lr of M-tile t.  Tiles are TR x TC blocks of the band's RB x WO map
    (block-row-major), or - when an image has fewer than 16 positions - consecutive whole images."""
    WO, TC = g["WO"], g["TC"]
    if g["PI"] >= 16:
        TR = 16 // TC
        tiles_per_row = WO // TC
        per_img = g["PI"] // 16 if g["multi"] else None
        il = t // per_img if g["multi"] else 0
        tt = t % per_img if g["multi"] else t
        ty, tx = tt // tiles_per_row, tt % tiles_per_row
        return il, ty * TR + lr // TC, tx * TC + lr % TC
    ipt = 16 // g["PI"]                       # images per tile
    il = t * ipt + lr // g["PI"]
    q = lr % g["PI"]
    return il, q // WO, q % WO


def posoff(g, RS, ISZ, t, lr):
    il, oy, ox = tile_pos(g, t, lr)
    return il * ISZ + g["S"] * oy * RS + g["S"] * ox


def ok(g, RS, ISZ, PS):
    for t in range(g["BPOS"] // 16):
        for half in (0, 2):
            banks = {(lq * PS + posoff(g, RS, ISZ, t, lr)) % 32 for lq in (half, half + 1) for lr in range(16)}
            if len(banks) != 32:
                return False
    return True


def search(HIN, S, limit=80 * 1024, kind=0, bpos_list=(64, 32, 16)):
    HO = HIN // S
    best = None
    for BPOS in bpos_list:
        if HO * HO >= 16 and (BPOS % HO and HO % 1 == 0) and BPOS < HO:
            continue
        tcs = [tc for tc in (16, 8, 4, 2) if tc <= HO and 16 % tc == 0] if HO * HO >= 16 else [HO]
        for TC in tcs:
            g = geo(HIN, S, BPOS, TC, kind)
            if g["PI"] >= 16:
                TR = 16 // TC
                if g["RB"] % TR or g["RB"] < TR or (not g["multi"] and (g["PI"] % BPOS or BPOS % (TR * g["WO"]))):
                    continue
            rs_min = HIN + ((2 if S == 1 else 1) if kind == 0 else 1)
            for RS in range(rs_min, rs_min + 33):
                found = None
                for ipad in range(0, 33 if g["multi"] else 1):
                    ISZ = g["PR"] * RS + ipad
                    for ppad in range(0, 33):
                        PS = g["NI"] * ISZ + ppad
                        if 64 * PS * 4 <= limit and ok(g, RS, ISZ, PS):
                            found = (PS, RS, ISZ)
                            break
                    if found:
                        break
                if found:
                    cand = (-BPOS, found[0], TC, found[1], found[2])
                    if best is None or cand < best[0]:
                        best = (cand, g)
                    break
        if best is not None and -best[0][0] == BPOS:
            break
    return best


def main_dgrad():
    print("stride-2 data gradient (dy map size HO; 4 classes x BPOS/16 accumulators per wave)")
    for HO in (32, 16, 8, 4, 2):
        (nb, PS, TC, RS, ISZ), g = search(HO, 1, kind=1, bpos_list=(32, 16))
        print(f"HO={HO:3d}: BPOS={-nb:2d} TC={TC:2d} NI={g['NI']:2d} RB={g['RB']:2d} PR={g['PR']:2d} RS={RS:3d} ISZ={ISZ:4d} PS={PS:4d}"
              f"  patch = {64 * PS * 4 / 1024:.1f} KiB")


if __name__ == "__main__":
    main_dgrad()
    for HIN, S in [(64, 2), (32, 1), (32, 2), (16, 1), (16, 2), (8, 1), (8, 2), (4, 1), (4, 2), (2, 1)]:
        (nb, PS, TC, RS, ISZ), g = search(HIN, S)
        print(f"HIN={HIN:3d} S={S}: BPOS={-nb:2d} TC={TC:2d} NI={g['NI']:2d} RB={g['RB']:2d} PR={g['PR']:2d} RS={RS:3d} ISZ={ISZ:4d} PS={PS:4d}"
              f"  patch = {64 * PS * 4 / 1024:.1f} KiB")


# ---- weight gradient: K = positions.  B operand = x slice [16 ci][img][row][col] (lane lr = channel, lq = which of the 4
# positions of the k-step), A operand = dy tile.  Positions of k-step ks: pb = blk*4Q + lq*Q + j, ks = blk*Q + j. -----------
def wgrad_search(HIN, S, BPOS):
    HO = HIN // S
    WO, PI = HO, HO * HO
    multi = PI < BPOS
    NI = BPOS // PI if multi else 1
    RB = HO if multi else BPOS // WO
    PR = S * (RB - 1) + 3
    rs_min = HIN + (2 if S == 1 else 1)
    best = None

    def posoff(pb, RS, ISZ):
        il = pb // PI if multi else 0
        q = pb % PI if multi else pb
        return il * ISZ + S * (q // WO) * RS + S * (q % WO)

    for Q in (1, 2, 4, 8, 16):
        if 4 * Q > BPOS:
            continue
        nks = BPOS // 4

        def pbs(ks):
            blk, j = ks // Q, ks % Q
            return [blk * 4 * Q + lq * Q + j for lq in range(4)]
        for RS in range(rs_min, rs_min + 33):
            done = False
            for ipad in range(0, 33 if multi else 1):
                ISZ = PR * RS + ipad
                for ppad in range(0, 33):
                    PS = NI * ISZ + ppad
                    good = True
                    for ks in range(nks):
                        pb = pbs(ks)
                        for half in (0, 2):
                            banks = {(lr * PS + posoff(pb[lq], RS, ISZ)) % 32 for lq in (half, half + 1) for lr in range(16)}
                            if len(banks) != 32:
                                good = False
                                break
                        if not good:
                            break
                    if good:
                        # dy tile: transposed [pos][co] with stride DS (lanes: lr = co), or natural [co][pos]
                        dsT = next((DS for DS in range(64, 100) if all(len({(pbs(ks)[lq] * DS + lr) % 32 for lq in (h, h + 1) for lr in range(16)}) == 32
                                                                           for ks in range(nks) for h in (0, 2))), None)
                        dsN = next((DS for DS in range(BPOS, BPOS + 40) if all(len({(lr * DS + pbs(ks)[lq]) % 32 for lq in (h, h + 1) for lr in range(16)}) == 32
                                                                                 for ks in range(nks) for h in (0, 2))), None)
                        cand = (16 * PS, Q, RS, ISZ, PS, dsT, dsN)
                        if (dsT or dsN) and (best is None or cand < best):
                            best = cand
                        done = True
                        break
                if done:
                    break
            if done and best is not None and best[1] == Q:
                break
    return dict(NI=NI, RB=RB, PR=PR), best


def main_wgrad():
    print("weight gradient (x slice of 16 channels + dy tile)")
    for HIN, S, BPOS in [(64, 2, 128), (32, 1, 128), (32, 2, 128), (16, 1, 128), (16, 2, 128), (8, 1, 128), (8, 2, 128), (4, 1, 128), (4, 2, 128), (2, 1, 128)]:
        g, best = wgrad_search(HIN, S, BPOS)
        if best is None:
            print(HIN, S, "none")
            continue
        fl, Q, RS, ISZ, PS, dsT, dsN = best
        print(f"HIN={HIN:3d} S={S} BPOS={BPOS}: Q={Q:2d} NI={g['NI']:2d} PR={g['PR']:2d} RS={RS:3d} ISZ={ISZ:4d} PS={PS:4d} dsT={dsT} dsN={dsN}"
              f"  x slice = {fl * 4 / 1024:.1f} KiB, dy tile = {BPOS * 64 * 4 / 1024:.0f} KiB")


if __name__ == "__main__":
    main_wgrad()
