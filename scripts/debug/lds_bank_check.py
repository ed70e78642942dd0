# brute-force LDS bank-conflict check for the dual-use weight image and the dW exchange image
from collections import defaultdict
def conflicts(groups, addr_fn, nbytes, mod):
    worst = 0
    for grp in groups:
        banks = defaultdict(set)
        for lane in grp:
            a = addr_fn(lane)
            for b in range(0, nbytes, 4):
                banks[((a + b) // 4) % mod].add((a + b) // 4)
        worst = max(worst, max(len(v) for v in banks.values()))
    return worst
G128 = [list(range(0,4))+list(range(12,16))+list(range(20,28)), list(range(4,12))+list(range(16,20))+list(range(28,32)),
        [32+x for x in list(range(0,4))+list(range(12,16))+list(range(20,28))], [32+x for x in list(range(4,12))+list(range(16,20))+list(range(28,32))]]
G64 = [list(range(32)), list(range(32, 64))]
GW64 = [list(range(16*k, 16*k+16)) for k in range(4)]

def f(row): return ((row & 3) << 2) | ((row >> 2) & 3)
def woff(row, ch): return 256 * row + 16 * (ch ^ f(row))
# forward row read: lane (i,h), tile t, k-step s, plane pl: chunk = 8*pl + 2*s + h
w = 0
for t in range(2):
    for s in range(4):
        for pl in range(2):
            w = max(w, conflicts(G128, lambda L: woff(32*t + (L & 31), 8*pl + 2*s + (L >> 5)), 16, 64))
print("weight image, forward ds_read_b128 worst:", w)
# transposed read: lane L = 16g+4q+p ; h=g>>1 ; row R = 16s+8jj+4h+q ; col c = 32t+16(g&1)+4p
def tr_addr(L, t, s, jj, pl):
    g, q, p = L >> 4, (L >> 2) & 3, L & 3
    h = g >> 1
    R = 16*s + 8*jj + 4*h + q
    c = 32*t + 16*(g & 1) + 4*p
    sc, a, hh = c >> 4, (c >> 3) & 1, (c >> 2) & 1
    run = 4*sc + 2*hh + a
    return woff(R, 8*pl + (run >> 1)) + 8*(run & 1)
w = 0
for t in range(2):
    for s in range(4):
        for jj in range(2):
            for pl in range(2):
                w = max(w, conflicts(G64, lambda L: tr_addr(L, t, s, jj, pl), 8, 64))
print("weight image, tr_b16 worst:", w)

# exchange image [32 particles][64 feat] f16, 128-B rows, 8-B units swizzled
def fx(j): return (j & 5) | ((j & 2) << 2) | ((j & 8) >> 2)
def xoff(p, unit): return 128 * p + 8 * (unit ^ fx(p & 15))
# writes: lane (j,h), tile t, g: features 32t+8g+4h.. -> unit = 8t+2g+h
w = 0
for t in range(2):
    for g in range(4):
        w = max(w, conflicts(GW64, lambda L: xoff(L & 31, 8*t + 2*g + (L >> 5)), 8, 32))
print("exchange image, ds_write_b64 worst:", w)
# tr reads: L=16g+4q+p; h'=g>>1; row P = 16kk+8h'+4jj+q; feature col 32mt+16(g&1)+4p -> unit 8mt+4(g&1)+p
w = 0
for mt in range(2):
    for kk in range(2):
        for jj in range(2):
            def ad(L):
                g, q, p = L >> 4, (L >> 2) & 3, L & 3
                P = 16*kk + 8*(g >> 1) + 4*jj + q
                return xoff(P, 8*mt + 4*(g & 1) + p)
            w = max(w, conflicts(G64, ad, 8, 64))
print("exchange image, tr_b16 worst:", w)

# ---- the closed forms the kernel uses (particle_net_train_fused.inc) against the definitions above
def swz(row): return ((row & 3) << 2) | ((row >> 2) & 3)
for L in range(64):
    i, h = L & 31, L >> 5
    g, q, p = L >> 4, (L >> 2) & 3, L & 3
    base_r = 256 * i + 16 * (h ^ swz(i))
    base_t = 256 * (4 * h + q) + 8 * (p >> 1) + 16 * ((2 * (g & 1) + (p & 1)) ^ h ^ (4 * q))
    for t in range(2):
        for s in range(4):
            for pl in range(2):
                assert (base_r ^ (128 * pl + 32 * s)) + 8192 * t == woff(32 * t + i, 8 * pl + 2 * s + h)
                for jj in range(2):
                    assert (base_t ^ (128 * pl + 64 * t + 32 * jj)) + 2048 * jj + 4096 * s == tr_addr(L, t, s, jj, pl)
    j = L & 31
    wx = 128 * j + 8 * (fx(j) ^ h)
    for t in range(2):
        for gg in range(4):
            assert wx ^ (64 * t + 16 * gg) == xoff(j, 8 * t + 2 * gg + h)
    base_rx = 128 * (8 * h + q) + 8 * ((4 * (g & 1) + p) ^ ((q & 1) | (h << 1) | ((q >> 1) << 3)))
    for mt in range(2):
        for kk in range(2):
            for jj in range(2):
                P = 16 * kk + 8 * h + 4 * jj + q
                assert (base_rx ^ (64 * mt) ^ (32 * jj)) + 512 * jj + 2048 * kk == xoff(P, 8 * mt + 4 * (g & 1) + p)
print("closed-form addresses agree with the definitions")
