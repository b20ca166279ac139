"""LDS bank-conflict model of MI355X_MICROARCH.md (LDS table) for the access patterns of the GEMM epilogue transpose.
A wave instruction is serviced in fixed lane groups, one LDS cycle per group when all dwords of the group fall into
distinct banks; each extra distinct address on a busy bank adds a cycle.  usage: lds_conflicts.py"""
import itertools

B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS = B128_GROUPS + [[l + 32 for l in g] for g in B128_GROUPS]
W64_GROUPS = [list(range(16 * g, 16 * g + 16)) for g in range(4)]
W128_GROUPS = [list(range(8 * g, 8 * g + 8)) for g in range(8)]


def cycles(addrs, groups, dwords, nbanks):
    """addrs: byte address per lane; returns (LDS cycles, conflict-free cycles)."""
    tot = 0
    for g in groups:
        banks = {}
        for l in g:
            for d in range(dwords):
                a = addrs[l] // 4 + d
                banks.setdefault(a % nbanks, set()).add(a)
        tot += max(len(v) for v in banks.values())
    return tot, len(groups)


def epilogue(MI, ES, pad, swz=None):
    """the fat epilogue's transpose region: write (r16 = pixel, q*4 channels of fragment i), read back 16-byte chunks
    pixel-major.  swz(pix, chunk) -> chunk: optional XOR of the 16-byte chunk index inside a pixel's row."""
    CHB = MI * 16 * ES
    ROWB = CHB + pad
    CPP = CHB // 16
    wr = rd = wr0 = rd0 = 0
    # writes: one instruction per fragment i; bf16: ds_write_b64 (4 channels = 8 B), f32: ds_write_b128 (16 B)
    for i in range(MI):
        addrs = []
        for l in range(64):
            r16, q = l & 15, l >> 4
            byte = i * 16 * ES + q * 4 * ES          # offset inside the pixel's row
            ch, within = byte // 16, byte % 16
            if swz:
                ch = swz(r16, ch)
            addrs.append(r16 * ROWB + ch * 16 + within)
        c, c0 = cycles(addrs, W64_GROUPS if ES == 2 else W128_GROUPS, 2 if ES == 2 else 4, 32)
        wr += c; wr0 += c0
    NST = 16 * CPP // 64
    for t in range(NST):
        addrs = []
        for l in range(64):
            ci = t * 64 + l
            pix, ch = ci // CPP, ci % CPP
            if swz:
                ch = swz(pix, ch)
            addrs.append(pix * ROWB + ch * 16)
        c, c0 = cycles(addrs, B128_GROUPS, 4, 64)
        rd += c; rd0 += c0
    return wr, wr0, rd, rd0


if __name__ == "__main__":
    for ES, MI in ((2, 6), (2, 4), (2, 2), (4, 6), (4, 4), (4, 2), (4, 1)):
        print(f"ES {ES} MI {MI} (row {MI * 16 * ES} B):")
        for pad in range(0, 80, 8 if ES == 2 else 16):
            if (MI * 16 * ES + pad) % 8:
                continue
            w, w0, r, r0 = epilogue(MI, ES, pad)
            tag = "  <- current" if pad == 16 else ""
            al = "" if (MI * 16 * ES + pad) % 16 == 0 else " (rows 8-B aligned only)"
            print(f"   pad {pad:3d}: write {w:3d}/{w0:3d} cycles, read {r:3d}/{r0:3d}{al}{tag}")


def epilogue_rd64(MI, ES, pad):
    """same region, read back as two ds_read_b64 per 16-byte chunk (rows then need 8-byte alignment only)."""
    CHB = MI * 16 * ES; ROWB = CHB + pad; CPP = CHB // 16
    G64 = [list(range(32)), list(range(32, 64))]
    rd = rd0 = 0
    for t in range(16 * CPP // 64):
        for half in (0, 8):
            addrs = []
            for l in range(64):
                ci = t * 64 + l
                addrs.append((ci // CPP) * ROWB + (ci % CPP) * 16 + half)
            c, c0 = cycles(addrs, G64, 2, 64)
            rd += c; rd0 += c0
    return rd, rd0


if __name__ == "__main__":
    print("two ds_read_b64 per chunk:")
    for ES, MI in ((2, 6), (2, 4), (2, 2)):
        for pad in (8, 24, 40, 56):
            print(f"   ES {ES} MI {MI} pad {pad}: read {epilogue_rd64(MI, ES, pad)}  write {epilogue(MI, ES, pad)[:2]}")


def epilogue_swz(MI, ES, pad, s, m, fb):
    """chunk' = chunk ^ ((pix >> s) & m); fb >= 0: the two 8-byte halves of a chunk swapped where bit fb of pix is set
    (bf16 only: the writer's granule is 8 bytes)."""
    CHB = MI * 16 * ES; ROWB = CHB + pad; CPP = CHB // 16
    f = lambda pix: (pix >> s) & m
    wr = wr0 = rd = rd0 = 0
    for i in range(MI):
        addrs = []
        for l in range(64):
            r16, q = l & 15, l >> 4
            byte = i * 16 * ES + q * 4 * ES
            ch, within = (byte // 16) ^ f(r16), byte % 16
            if ch >= CPP:
                return None
            if ES == 2 and fb >= 0 and (r16 >> fb) & 1:
                within ^= 8
            addrs.append(r16 * ROWB + ch * 16 + within)
        c, c0 = cycles(addrs, W64_GROUPS if ES == 2 else W128_GROUPS, 2 if ES == 2 else 4, 32)
        wr += c; wr0 += c0
    for t in range(16 * CPP // 64):
        addrs = []
        for l in range(64):
            ci = t * 64 + l
            pix, ch = ci // CPP, (ci % CPP) ^ f(ci // CPP)
            addrs.append(pix * ROWB + ch * 16)
        c, c0 = cycles(addrs, B128_GROUPS, 4, 64)
        rd += c; rd0 += c0
    return wr, wr0, rd, rd0


if __name__ == "__main__":
    print("swizzle search (pad, shift, mask, half-flip bit): write, read cycles")
    for ES, MI in ((2, 6), (2, 4), (2, 2), (4, 6), (4, 4), (4, 2)):
        best = []
        for pad, s, m, fb in itertools.product((0, 16, 32, 48), range(4), (0, 1, 3, 7), (-1, 0, 1, 2, 3)):
            if ES == 4 and fb >= 0:
                continue
            r = epilogue_swz(MI, ES, pad, s, m, fb)
            if r:
                best.append((r[0] + r[2], pad, s, m, fb, r))
        best.sort()
        print(f"  ES {ES} MI {MI}: ideal {best[0][5][1] + best[0][5][3]}; best:", best[:4])
