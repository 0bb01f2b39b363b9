#!/usr/bin/env python3
"""Search the shape table of the compile-time-shaped filterbank kernels (pfb_mid.hip, WH_MID_CONFIGS): for a channel count
M, enumerate (R runs per workgroup, GH hops per group, image padding) and score each by lane utilisation, LDS footprint
and simulated LDS bank conflicts of the arm-MAC writes, the in-place passes and the last pass (bank rules of
MI355X_MICROARCH.md: ds_read_b64 = 2 groups of 32 lanes over 64 banks, ds_write_b64 = 4 groups of 16 lanes over 32 banks).
Mirrors MidCfg / pass_map / pos_of of pfb_mid.hip.   pfb_mid_configs.py [--r8] M[:R] [M[:R] ...]"""
import sys, itertools


R8 = "--r8" in sys.argv      # radix-8 passes for the power-of-two part (MidCfg's R8_ parameter)


def plan(Q):
    r, L, rem, cur = [], [], Q, Q
    if R8:
        k, q = 0, Q
        while q % 2 == 0: q //= 2; k += 1
        for _ in range(k // 3 - 1 if k % 3 == 1 else k // 3):
            r.append(8); L.append(cur); cur //= 8; rem //= 8
    for rad in (4, 2, 3, 5):
        while rem > 1 and rem % rad == 0:
            r.append(rad); L.append(cur); cur //= rad; rem //= rad
    return (r, L) if rem == 1 and r else None


def pass_map(LS, nblk, m):
    for ul in range(min(m, LS), 0, -1):
        if m % ul == 0 and LS % ul == 0 and nblk % (LS // ul) == 0:
            return ul, LS // ul, m // ul, nblk // (LS // ul)
    return None


def rd_cycles(addrs):   # ds_read_b64
    cyc = 0
    for g in (range(0, 32), range(32, 64)):
        banks = {}
        for l in g:
            if l < len(addrs) and addrs[l] is not None:
                for d in (2 * addrs[l], 2 * addrs[l] + 1):
                    banks.setdefault(d % 64, set()).add(d)
        cyc += max([len(v) for v in banks.values()] or [1])
    return cyc


def wr_cycles(addrs):   # ds_write_b64
    cyc = 0
    for g0 in range(0, 64, 16):
        banks = {}
        for l in range(g0, g0 + 16):
            if l < len(addrs) and addrs[l] is not None:
                for d in (2 * addrs[l], 2 * addrs[l] + 1):
                    banks.setdefault(d % 32, set()).add(d)
        cyc += max([len(v) for v in banks.values()] or [1])
    return cyc


def evaluate(M, R, GH, PB, PADN, IMGX, LSP=0, T=9):
    Q = M // 4
    pl = plan(Q)
    if not pl: return None
    rad, Ls = pl
    NT = R * Q
    NW = (NT + 63) // 64
    NTL = NW * 64
    NIMG = R * GH
    if NTL > 1024: return None
    if PB and Q % PB: return None
    ph = (lambda p: p + (p // PB) * PADN) if PB else (lambda p: p)
    IMGS = ph(M) + IMGX
    LS = LSP if LSP else NT // NIMG
    TH = LS * NIMG
    if TH == 0 or TH > NT: return None
    np_ = len(rad)
    tot_r = ideal_r = tot_w = ideal_w = 0
    # passes 0..np-2 (workgroup mode: thread lt -> hop lt // LS)
    for p in range(np_ - 1):
        r, L = rad[p], Ls[p]; m = L // r
        pm = pass_map(LS, M // L, m)
        if not pm: return None
        UL, BL, CU, CB = pm
        for ls in range(LS):     # affine check
            bl, ul = divmod(ls, UL)
            for cb in range(CB):
                for cu in range(CU):
                    for j in range(r):
                        if ph((bl + BL * cb) * L + ul + UL * cu + j * m) != ph(bl * L + ul) + ph(BL * cb * L + UL * cu) + ph(j * m):
                            return None
        for w0 in range(0, TH, 64):
            for cb in range(CB):
                for cu in range(CU):
                    for j in range(r):
                        ad = []
                        for lt in range(w0, min(w0 + 64, TH)):
                            hopl, ls = divmod(lt, LS); bl, ul = divmod(ls, UL)
                            ad.append(hopl * IMGS + ph((bl + BL * cb) * L + ul + UL * cu + j * m))
                        tot_r += rd_cycles(ad); ideal_r += 2
                        tot_w += wr_cycles(ad); ideal_w += 4
    # last pass: lane <-> kb
    RL = rad[-1]; BPL = M // RL
    def pos_of(kb):
        rem, pos = kb >> 2, (kb & 3) * Q
        for p in range(np_ - 1):
            pos += (rem % rad[p]) * (Ls[p] // rad[p]); rem //= rad[p]
        return pos

    def lsl_ok(c):
        if BPL % c or NIMG % (64 // c) or ((NIMG // (64 // c)) * (BPL // c)) % NW: return False
        return all(ph(pos_of(kl + c * cc) + j) == ph(pos_of(kl)) + ph(pos_of(c * cc)) + j
                   for kl in range(c) for cc in range(BPL // c) for j in range(RL))
    LSL = next((c for c in (64, 32, 16, 8, 4) if lsl_ok(c)), None)
    if LSL is None: return None

    HR = 64 // LSL
    for c in range(BPL // LSL):
        for j in range(RL):
            ad = []
            for lane in range(64):
                hs, kl = divmod(lane, LSL)
                ad.append(hs * IMGS + ph(pos_of(kl + LSL * c)) + j)
            tot_r += rd_cycles(ad) * (NIMG // HR); ideal_r += 2 * (NIMG // HR)
    # arm-MAC writes
    for w0 in range(0, NT, 64):
        for i in range(GH):
            for k1 in range(4):
                ad = []
                for tid in range(w0, min(w0 + 64, NT)):
                    r_, u = divmod(tid, Q)
                    ad.append((r_ * GH + i) * IMGS + ph(k1 * Q + u))
                tot_w += wr_cycles(ad); ideal_w += 4
    lds = NIMG * IMGS * 8 + sum((rad[p] - 1) * (Ls[p] // rad[p]) for p in range(np_ - 1)) * 16
    util = NT / NTL
    putil = TH / NTL
    return dict(M=M, R=R, GH=GH, PB=PB, PADN=PADN, IMGX=IMGX, NW=NW, LS=LS, lds=lds, util=util, rd=tot_r / ideal_r, wr=tot_w / ideal_w,
                plan=rad, LSL=LSL, LSP=LSP, putil=putil)


def best(M, force_R=None):
    Q = M // 4
    pl = plan(Q)
    if not pl: return None
    rad, Ls = pl
    cands = []
    pbs = sorted({0, Q} | set(Ls[1:]))
    for R in ([force_R] if force_R else range(1, 17)):
        NT = R * Q
        NW = (NT + 63) // 64
        if NW > 16 or (not force_R and NT / (NW * 64) < 0.88): continue
        for GH in (4, 3, 2, 5, 6, 8, 1):
            for PB in pbs:
                for PADN in ((0,) if PB == 0 else (1, 2)):
                    for IMGX in range(0, 5):
                        full = NT // (R * GH)
                        for LSP in [0] + [l for l in range(full - 1, full * 5 // 8, -1)]:
                            e = evaluate(M, R, GH, PB, PADN, IMGX, LSP)
                            if e and e["lds"] <= 80 * 1024:
                                cands.append(e)
                                break
    if not cands: return None
    # 4-wave workgroups first (one wave per SIMD), then few conflicts, then small LDS
    cands.sort(key=lambda e: (e["NW"] % 4 != 0, e["NW"] > 8, e["putil"] < 0.8, abs(e["GH"] - 4), round(e["rd"] + 1.5 * e["wr"] - e["putil"], 2), e["lds"]))
    return cands[0]


if __name__ == "__main__":
    for arg in (a for a in sys.argv[1:] if not a.startswith("--")):
        M, fr = (int(v) for v in arg.split(":")) if ":" in arg else (int(arg), None)
        b = best(M, fr)
        if not b: print(f"M={M}: no workgroup-mode shape"); continue
        wpe = 3 if b["NW"] in (3, 4, 12) else 2
        r8s = ", 1" if R8 else ""
        print(f"    X({M}, {b['R']}, {b['GH']}, 0, {wpe}, {b['PB']}, {b['PADN']}, {b['IMGX']}, 36, {b['LSP']}{r8s}) \\   // plan {b['plan']} waves {b['NW']} lanes/hop {b['LS']} "
              f"util {b['util']:.2f} pass-util {b['putil']:.2f} last-pass lanes/hop {b['LSL']} LDS {b['lds']} B conflicts rd x{b['rd']:.2f} wr x{b['wr']:.2f}")
