"""LDS bank-conflict model for gfx950 (guides/MI355X_MICROARCH.md, LDS table): per-instruction lane groups and
bank function.  Used to choose / verify the swizzles of the GEMM and attention LDS images."""
import itertools

B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
               list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
               list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
HALF_GROUPS = [list(range(0, 32)), list(range(32, 64))]


def conflicts(addrs, width, groups, nbanks=64):
    """max number of distinct addresses hitting one bank within a lane group (1 = conflict-free)."""
    worst = 1
    for grp in groups:
        bank_addrs = {}
        for l in grp:
            a = addrs[l]
            for d in range(width // 4):
                bank_addrs.setdefault(((a // 4) + d) % nbanks, set()).add(a)
        worst = max(worst, max(len(v) for v in bank_addrs.values()))
    return worst


def b128(addrs):
    return conflicts(addrs, 16, B128_GROUPS)


def tr_b64(addrs):
    return conflicts(addrs, 8, HALF_GROUPS)


if __name__ == '__main__':
    # ---- 128-B rows (BK=64), chunk' = chunk ^ (row & 7): fragment reads of gemm_tn / attention
    for ks in range(2):
        addrs = [(l & 15) * 128 + (((ks * 4 + (l >> 4)) ^ (l & 7)) << 4) for l in range(64)]
        print('128B rows b128 ks', ks, 'ways', b128(addrs))
    # transposing reads of 8 consecutive rows on the same image (attention V / Q / dO / K tiles)
    for s in range(2):
        for cb in range(4):
            for hi in range(2):
                addrs = []
                for l in range(64):
                    i, g = l & 15, l >> 4
                    row = 32 * s + 4 * g + (i >> 2) + 16 * hi
                    p = i & 3
                    lch = 2 * cb + (p >> 1)
                    addrs.append(row * 128 + ((lch ^ (row & 7)) << 4) + (p & 1) * 8)
                w = tr_b64(addrs)
                assert w == 1, (s, cb, hi, w)
    print('128B rows tr_b16 reads: conflict-free')
    # wgrad image: 256-B rows, 32-B slots, slot' = slot ^ h(row)
    for ks in range(2):
        for blk in range(8):
            for hi in range(2):
                addrs = []
                for l in range(64):
                    g, li = l >> 4, l & 15
                    r = ks * 32 + 8 * g + (li >> 2) + 4 * hi
                    h = (r & 3) | (((r >> 3) & 1) << 2)
                    addrs.append(r * 256 + ((blk ^ h) << 5) + (li & 3) * 8)
                assert tr_b64(addrs) == 1
    print('wgrad 256B rows tr_b16 reads: conflict-free')
    # ---- search: 64-B rows (BK=32): chunk' = chunk ^ f(row), f linear over GF(2) in the 4 low row bits
    best = []
    for m in itertools.product(range(16), repeat=2):        # two output bits, each a mask over row bits
        def f(row, m=m):
            return (bin(row & m[0]).count('1') & 1) | ((bin(row & m[1]).count('1') & 1) << 1)
        addrs = [(l & 15) * 64 + (((l >> 4) ^ f(l & 15)) << 4) for l in range(64)]
        w = b128(addrs)
        if w == 1:
            best.append(m)
    print('64B rows: conflict-free linear swizzles (bit0 mask, bit1 mask):', best[:10], len(best))
