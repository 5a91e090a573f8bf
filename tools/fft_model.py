#!/usr/bin/env python3
"""fft_model.py — numpy model of the wave-private 4096-point FFT data movement used by csrc/if_fir_fft.hip.

One wave = 64 lanes x 64 registers = one 4096-point complex FFT, N = 16*16*16:
    n = 256*n0 + 16*n1 + n2   (input),      k = k0 + 16*k1 + 256*k2   (output)
  load      reg[row] of lane = x[64*row + lane]            -> n0 = row//4, rho = row%4, n1 = lane//16 + 4*rho, n2 = lane%16
  pass 1    FFT16 over n0 (4 groups rho)                   -> slot (k0, rho);  twiddle W4096^((lane+64*rho)*k0)
  exch 1    permlane32_swap + permlane16_swap              -> lane g=lane//16 keeps k0 = 4g+i; slot (i,c,b,rho), n1 = c+2b+4rho
  pass 2    FFT16 over n1 (4 groups i)                     -> slot (i, k1);    twiddle W256^(n2*k1), n2 = lane%16
  exch 2    16x16 transposition inside each 16-lane row (LDS) -> lane%16 = k1, slot (i, n2)
  pass 3    FFT16 over n2                                  -> slot (i, k2):  X[(4g+i) + 16*(lane%16) + 256*k2]
The inverse runs the mirror image and ends in the load layout.  This file checks the index algebra against numpy.fft
and is the reference the HIP kernel was transcribed from (development tool; not product, not oracle)."""
import numpy as np

N = 4096
W = lambda n, e: np.exp(-2j * np.pi * e / n)  # noqa: E731


def swap32(vdst, src):
    """v_permlane32_swap: lanes 32-63 of vdst <-> lanes 0-31 of src."""
    a, b = vdst.copy(), src.copy()
    a[32:], b[:32] = src[:32].copy(), vdst[32:].copy()
    return a, b


def swap16(vdst, src):
    """v_permlane16_swap: odd 16-lane rows of vdst <-> even rows of src."""
    a, b = vdst.copy(), src.copy()
    for row in (0, 2):
        a[16 * (row + 1):16 * (row + 2)] = src[16 * row:16 * (row + 1)]
        b[16 * row:16 * (row + 1)] = vdst[16 * (row + 1):16 * (row + 2)]
    return a, b


def forward(x):
    lane = np.arange(64)
    reg = np.zeros((64, 64), dtype=np.complex128)       # reg[slot, lane]
    for row in range(64):
        reg[row] = x[64 * row + lane]
    # pass 1: FFT16 over n0 for each rho; slot (k0, rho) := 4*k0 + rho
    p1 = np.zeros_like(reg)
    for rho in range(4):
        a = np.fft.fft(reg[rho::4], axis=0)              # rows 4*n0+rho, n0 = 0..15
        for k0 in range(16):
            p1[4 * k0 + rho] = a[k0] * W(N, (lane + 64 * rho) * k0)
    # exchange 1, stage A: pair slot (k0<8, rho) with (k0+8, rho), permlane32_swap(vdst = k0<8, src = k0+8)
    e = p1.copy()
    for k0 in range(8):
        for rho in range(4):
            e[4 * k0 + rho], e[4 * (k0 + 8) + rho] = swap32(e[4 * k0 + rho], e[4 * (k0 + 8) + rho])
    # stage B: k0slot = i + 4c + 8b; pair (i, c=0, b, rho) with (i, c=1, b, rho), permlane16_swap(vdst = c0, src = c1)
    for i in range(4):
        for b in range(2):
            for rho in range(4):
                s0, s1 = 4 * (i + 8 * b) + rho, 4 * (i + 4 + 8 * b) + rho
                e[s0], e[s1] = swap16(e[s0], e[s1])
    # now lane g holds k0 = 4g+i; slot (i,c,b,rho) holds n1 = c + 2b + 4rho
    p2 = np.zeros_like(reg)                              # slot (i, k1) := 16*i + k1
    n2 = lane % 16
    for i in range(4):
        seq = np.zeros((16, 64), dtype=np.complex128)
        for c in range(2):
            for b in range(2):
                for rho in range(4):
                    seq[c + 2 * b + 4 * rho] = e[4 * (i + 4 * c + 8 * b) + rho]
        a = np.fft.fft(seq, axis=0)
        for k1 in range(16):
            p2[16 * i + k1] = a[k1] * W(256, n2 * k1)
    # exchange 2: inside each 16-lane row g: element (i, k1) at lane (g, n2) -> lane (g, k1), slot (i, n2)
    e2 = np.zeros_like(reg)
    for i in range(4):
        for g in range(4):
            blk = p2[16 * i:16 * i + 16, 16 * g:16 * g + 16]   # [k1, n2]
            e2[16 * i:16 * i + 16, 16 * g:16 * g + 16] = blk.T  # [n2, k1]
    # pass 3: FFT16 over n2 -> slot (i, k2)
    p3 = np.zeros_like(reg)
    for i in range(4):
        p3[16 * i:16 * i + 16] = np.fft.fft(e2[16 * i:16 * i + 16], axis=0)
    return p3


def k_of(slot, lane):
    i, k2 = slot // 16, slot % 16
    return (4 * (lane // 16) + i) + 16 * (lane % 16) + 256 * k2


def inverse(p3):
    lane = np.arange(64)
    n2 = lane % 16
    # pass 3^-1
    e2 = np.zeros_like(p3)
    for i in range(4):
        e2[16 * i:16 * i + 16] = np.fft.ifft(p3[16 * i:16 * i + 16], axis=0) * 16
    # exchange 2 (same transposition), now element (i, n2) at lane (g,k1) -> lane (g,n2), slot (i,k1)
    p2 = np.zeros_like(p3)
    for i in range(4):
        for g in range(4):
            p2[16 * i:16 * i + 16, 16 * g:16 * g + 16] = e2[16 * i:16 * i + 16, 16 * g:16 * g + 16].T
    e = np.zeros_like(p3)
    for i in range(4):
        seq = np.stack([p2[16 * i + k1] * np.conj(W(256, n2 * k1)) for k1 in range(16)])
        a = np.fft.ifft(seq, axis=0) * 16                   # index n1
        for c in range(2):
            for b in range(2):
                for rho in range(4):
                    e[4 * (i + 4 * c + 8 * b) + rho] = a[c + 2 * b + 4 * rho]
    # exchange 1 inverse: stage B then stage A (each an involution with the same pairing)
    for i in range(4):
        for b in range(2):
            for rho in range(4):
                s0, s1 = 4 * (i + 8 * b) + rho, 4 * (i + 4 + 8 * b) + rho
                e[s0], e[s1] = swap16(e[s0], e[s1])
    for k0 in range(8):
        for rho in range(4):
            e[4 * k0 + rho], e[4 * (k0 + 8) + rho] = swap32(e[4 * k0 + rho], e[4 * (k0 + 8) + rho])
    out = np.zeros_like(p3)
    for rho in range(4):
        seq = np.stack([e[4 * k0 + rho] * np.conj(W(N, (lane + 64 * rho) * k0)) for k0 in range(16)])
        a = np.fft.ifft(seq, axis=0) * 16                   # index n0
        for n0 in range(16):
            out[4 * n0 + rho] = a[n0]
    y = np.zeros(N, dtype=np.complex128)
    for row in range(64):
        y[64 * row + lane] = out[row]
    return y / N


def main():
    rng = np.random.default_rng(1)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    p3 = forward(x)
    ref = np.fft.fft(x)
    got = np.zeros(N, dtype=np.complex128)
    for slot in range(64):
        got[k_of(slot, np.arange(64))] = p3[slot]
    print("forward max err", np.max(np.abs(got - ref)))
    y = inverse(p3)
    print("roundtrip max err", np.max(np.abs(y - x)))
    # overlap-save filtering check
    h = rng.standard_normal(1023)
    H = np.fft.fft(h, N)
    Hp = np.zeros((64, 64), dtype=np.complex128)
    for slot in range(64):
        Hp[slot] = H[k_of(slot, np.arange(64))]
    yb = inverse(p3 * Hp)
    full = np.convolve(x, h)[:N]
    print("overlap-save valid-part err", np.max(np.abs(yb[1022:] - full[1022:])))


if __name__ == "__main__":
    main()


# ---------------------------------------------------------------------------------------------------------------
# decimate-by-4 variant: fold the 4 aliases in the frequency domain (in-lane) and run a 1024-point inverse
#   k' = k0 + 16*k1 + 256*k2'   (k0 = 4g+i, k1 = lane%16, k2' in 0..3)      m' = 64*mu0 + 4*mu1 + mu2
#   A: 4-point iDFT over k2' -> slot (i, mu2); twiddle conj W1024^((16*k1 + k0)*mu2)
#   X: row transposition: element j=4i+mu2 of lane (g,k1) -> lane (g,j), slot k1
#   B: iFFT16 over k1 -> mu1; twiddle conj W256^(k0*mu1), k0 = 4g + (lane%16)//4
#   Y: transposition among the 16 lanes sharing mu2: element mu1 of lane (k0, mu2) -> lane 4*mu1+mu2, slot k0
#   C: iFFT16 over k0 -> mu0.   result: lane = 4*mu1+mu2, slot mu0:  y_D[64*mu0 + lane]
def inverse_dec4(p3):
    lane = np.arange(64)
    g, m = lane // 16, lane % 16
    z = np.zeros((16, 64), dtype=np.complex128)          # slot (i, k2') := 4*i + k2'
    for i in range(4):
        for k2p in range(4):
            z[4 * i + k2p] = sum(p3[16 * i + k2p + 4 * j] for j in range(4))
    a = np.zeros_like(z)                                  # slot j = 4*i + mu2
    for i in range(4):
        seq = np.fft.ifft(z[4 * i:4 * i + 4], axis=0) * 4
        for mu2 in range(4):
            a[4 * i + mu2] = seq[mu2] * np.conj(W(1024, (16 * m + 4 * g + i) * mu2))
    x = np.zeros_like(a)                                  # lane (g, j), slot k1
    for gg in range(4):
        x[:, 16 * gg:16 * gg + 16] = a[:, 16 * gg:16 * gg + 16].T
    j = lane % 16
    i_l, mu2_l = j // 4, j % 4
    k0_l = 4 * g + i_l
    b = np.fft.ifft(x, axis=0) * 16                       # slot mu1
    for mu1 in range(16):
        b[mu1] = b[mu1] * np.conj(W(256, k0_l * mu1))
    y = np.zeros_like(b)                                  # lane 4*mu1+mu2, slot k0
    for src_lane in range(64):
        for mu1 in range(16):
            y[k0_l[src_lane], 4 * mu1 + mu2_l[src_lane]] = b[mu1, src_lane]
    c = np.fft.ifft(y, axis=0) * 16                       # slot mu0
    out = np.zeros(1024, dtype=np.complex128)
    for mu0 in range(16):
        out[64 * mu0 + lane] = c[mu0]
    return out / N


def main_dec4():
    rng = np.random.default_rng(2)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    h = rng.standard_normal(255)
    H = np.fft.fft(h, N)
    Hp = np.zeros((64, 64), dtype=np.complex128)
    for slot in range(64):
        Hp[slot] = H[k_of(slot, np.arange(64))]
    yd = inverse_dec4(forward(x) * Hp)
    full = np.convolve(x, h)[:N]
    print("dec4 valid-part err", np.max(np.abs(yd[64:] - full[256::4])))


if __name__ == "__main__":
    main_dec4()


# ---------------------------------------------------------------------------------------------------------------
# 16-slot filter bank at the channel rate (decimation 16, round 3): all 16 slots from ONE forward transform.
#   after pass 2 + exchange 2: lane (g, k1), group i, slot n2: t[n2]   (k0 = 4g + i)
#   channel s = prototype H moved up by s slots (s * 256 bins): H_s(k2) = H((k2 - s) mod 16) along the k2 axis
#   folded spectrum Z_s(k0, k1) = sum_k2 H_s(k2) Y(k2) = sum_n2 t[n2] G0[n2] W16^(n2 s) = FFT16(t * G0)[s],
#       G0[n2] = sum_k2 H(k0 + 16 k1 + 256 k2) W16^(n2 k2)                         (host table, same size as H)
#   256-point inverse per channel, 4 channels (cs) at a time = inverse_dec4 without its first stage:
#   X: element j = 4i + cs of lane (g, k1) -> lane (g, j), slot k1;   iFFT16 over k1 -> mu1;  twiddle conj W256^(k0 mu1)
#   Y: element mu1 of lane (k0, cs) -> lane 4 mu1 + cs, slot k0;      iFFT16 over k0 -> mu0
#   result: lane = 4 mu1 + cs, slot mu0:  y_s[16 mu0 + mu1]
def bank16(x, h):
    lane = np.arange(64)
    g, m = lane // 16, lane % 16
    H = np.fft.fft(h, N)
    # forward up to exchange 2 (copy of forward() without pass 3)
    p3 = forward(x)                                       # slot (i, k2): Y(k0 + 16 k1 + 256 k2)
    t = np.zeros_like(p3)
    for i in range(4):
        t[16 * i:16 * i + 16] = np.fft.ifft(p3[16 * i:16 * i + 16], axis=0)        # back to t[n2] (model shortcut)
    G0 = np.zeros((64, 64), dtype=np.complex128)         # slot (i, n2)
    for i in range(4):
        for n2 in range(16):
            G0[16 * i + n2] = sum(H[(4 * g + i) + 16 * m + 256 * k2] * W(16, n2 * k2) for k2 in range(16))
    z = np.zeros_like(p3)                                 # slot (i, s)
    for i in range(4):
        z[16 * i:16 * i + 16] = np.fft.fft(t[16 * i:16 * i + 16] * G0[16 * i:16 * i + 16], axis=0)
    out = np.zeros((16, 256), dtype=np.complex128)
    for b in range(4):
        a = np.zeros((16, 64), dtype=np.complex128)       # slot j = 4 i + cs
        for i in range(4):
            for cs in range(4):
                a[4 * i + cs] = z[16 * i + 4 * b + cs]
        xx = np.zeros_like(a)
        for gg in range(4):
            xx[:, 16 * gg:16 * gg + 16] = a[:, 16 * gg:16 * gg + 16].T
        j = lane % 16
        i_l, cs_l = j // 4, j % 4
        k0_l = 4 * g + i_l
        bb = np.fft.ifft(xx, axis=0) * 16
        for mu1 in range(16):
            bb[mu1] = bb[mu1] * np.conj(W(256, k0_l * mu1))
        y = np.zeros_like(bb)
        for src in range(64):
            for mu1 in range(16):
                y[k0_l[src], 4 * mu1 + cs_l[src]] = bb[mu1, src]
        c = np.fft.ifft(y, axis=0) * 16
        for mu0 in range(16):
            for ln in range(64):
                out[4 * b + ln % 4, 16 * mu0 + ln // 4] = c[mu0, ln]
    return out / N


def main_bank16():
    rng = np.random.default_rng(3)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    h = rng.standard_normal(255)
    out = bank16(x, h)
    n = np.arange(255)
    worst = 0.0
    for s in range(16):
        hs = h * np.exp(2j * np.pi * s * n / 16.0)        # prototype moved up by s/16 cycles/sample
        full = np.convolve(x, hs)[:N]
        worst = max(worst, np.max(np.abs(out[s][16:] - full[256::16])))
    print("bank16 valid-part err (all 16 slots)", worst)


if __name__ == "__main__":
    main_bank16()


# ---------------------------------------------------------------------------------------------------------------
# filter bank at decimation 8 (fs/16 slots, 2x oversampled channels, round 3): per channel, two channels per small inverse.
#   first radix-2 stage of pass 3 once:  w0[a] = t[a] + t[a+8],  w1[a] = t[a] - t[a+8]          (a = 0..7)
#   channel s, k2' in {0, 1}:  Z_s(k0, k1, k2') = sum_a w_{k2'}[a] * W16^(a s) * G_q[a],  q = (k2' - s) mod 2,
#       G_q[a] = W16^(a q) * sum_j H(k0 + 16 k1 + 256 (q + 2 j)) W8^(a j)                     (host table, same size as H)
#   512-point inverse (2 x 16 x 16), two channels at a time: stage A' = 2-point butterfly over k2' -> mu2, twiddle
#   conj W512^((16 k1 + k0) mu2), then the common tail with low = 2 ch + mu2:
#   result: lane = 4 mu1 + 2 ch + mu2, slot mu0:  y_ch[32 mu0 + 2 mu1 + mu2]
def bank8(x, h, slots):
    lane = np.arange(64)
    g, m = lane // 16, lane % 16
    H = np.fft.fft(h, N)
    p3 = forward(x)
    t = np.zeros_like(p3)
    for i in range(4):
        t[16 * i:16 * i + 16] = np.fft.ifft(p3[16 * i:16 * i + 16], axis=0)
    G = np.zeros((64, 64), dtype=np.complex128)          # slot (i, q, a) := 16 i + 8 q + a
    for i in range(4):
        for q in range(2):
            for a in range(8):
                G[16 * i + 8 * q + a] = W(16, a * q) * sum(H[(4 * g + i) + 16 * m + 256 * (q + 2 * j)] * W(8, a * j) for j in range(8))
    w = np.zeros_like(t)                                  # slot (i, k2', a) := 16 i + 8 k2' + a
    for i in range(4):
        for a in range(8):
            w[16 * i + a] = t[16 * i + a] + t[16 * i + a + 8]
            w[16 * i + 8 + a] = t[16 * i + a] - t[16 * i + a + 8]
    out = {}
    for pair in range(0, len(slots), 2):
        chans = slots[pair:pair + 2]
        a_ = np.zeros((16, 64), dtype=np.complex128)      # slot 4 i + 2 ch + mu2
        for ch, s in enumerate(chans):
            for i in range(4):
                z = []
                for k2p in range(2):
                    q = (k2p - s) % 2
                    z.append(sum(w[16 * i + 8 * k2p + a] * W(16, a * s) * G[16 * i + 8 * q + a] for a in range(8)))
                for mu2 in range(2):
                    v = z[0] + (-1) ** mu2 * z[1]
                    a_[4 * i + 2 * ch + mu2] = v * np.conj(W(512, (16 * m + 4 * g + i) * mu2))
        xx = np.zeros_like(a_)
        for gg in range(4):
            xx[:, 16 * gg:16 * gg + 16] = a_[:, 16 * gg:16 * gg + 16].T
        j = lane % 16
        i_l, low_l = j // 4, j % 4
        k0_l = 4 * g + i_l
        bb = np.fft.ifft(xx, axis=0) * 16
        for mu1 in range(16):
            bb[mu1] = bb[mu1] * np.conj(W(256, k0_l * mu1))
        y = np.zeros_like(bb)
        for src in range(64):
            for mu1 in range(16):
                y[k0_l[src], 4 * mu1 + low_l[src]] = bb[mu1, src]
        c = np.fft.ifft(y, axis=0) * 16
        for ch, s in enumerate(chans):
            o = np.zeros(512, dtype=np.complex128)
            for mu0 in range(16):
                for ln in range(64):
                    if (ln >> 1) & 1 == ch:
                        o[32 * mu0 + 2 * (ln >> 2) + (ln & 1)] = c[mu0, ln]
            out[s] = o / N
    return out


def main_bank8():
    rng = np.random.default_rng(4)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    h = rng.standard_normal(255)
    slots = [0, 1, 2, 5, 8, 11, 14, 15]
    out = bank8(x, h, slots)
    n = np.arange(255)
    worst = 0.0
    for s in slots:
        hs = h * np.exp(2j * np.pi * s * n / 16.0)
        full = np.convolve(x, hs)[:N]
        worst = max(worst, np.max(np.abs(out[s][32:] - full[256::8])))
    print("bank8 valid-part err", worst)


if __name__ == "__main__":
    main_bank8()


# ---------------------------------------------------------------------------------------------------------------
# decimation-8 bank, ALL SLOTS OF ONE PARITY (round 4): s = 2 sigma + par, W16^(a s) = W16^(a par) W8^(a sigma), so
#     Z_s(0) = FFT8_a( w0[a] . G_par[a] W16^(a par) )[sigma]        Z_s(1) = FFT8_a( w1[a] . G_(1-par)[a] W16^(a par) )[sigma]
# -- each of w0, w1 is used once: the block's 64 registers become the 64 values Z_s(k2') in place.  Checked against the
# per-channel sums of bank8 above.
def bank8_parity_z(x, h, par):
    lane = np.arange(64)
    g, m = lane // 16, lane % 16
    H = np.fft.fft(h, N)
    p3 = forward(x)
    t = np.zeros_like(p3)
    for i in range(4):
        t[16 * i:16 * i + 16] = np.fft.ifft(p3[16 * i:16 * i + 16], axis=0)
    G = np.zeros((64, 64), dtype=np.complex128)
    for i in range(4):
        for q in range(2):
            for a in range(8):
                G[16 * i + 8 * q + a] = W(16, a * q) * sum(H[(4 * g + i) + 16 * m + 256 * (q + 2 * j)] * W(8, a * j) for j in range(8))
    worst = 0.0
    for i in range(4):
        w0 = np.array([t[16 * i + a] + t[16 * i + a + 8] for a in range(8)])
        w1 = np.array([t[16 * i + a] - t[16 * i + a + 8] for a in range(8)])
        img0 = np.array([G[16 * i + 8 * par + a] * W(16, a * par) for a in range(8)])          # first half of the parity image
        img1 = np.array([G[16 * i + 8 * (1 - par) + a] * W(16, a * par) for a in range(8)])    # second half
        z0 = np.fft.fft(w0 * img0, axis=0)
        z1 = np.fft.fft(w1 * img1, axis=0)
        for sg in range(8):
            s = 2 * sg + par
            for k2p, z in ((0, z0), (1, z1)):
                q = (k2p - s) % 2
                ref = sum((w0, w1)[k2p][a] * W(16, a * s) * G[16 * i + 8 * q + a] for a in range(8))
                worst = max(worst, np.max(np.abs(ref - z[sg])))
    return worst


def main_bank8_parity():
    rng = np.random.default_rng(9)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    h = rng.standard_normal(255)
    for par in (0, 1):
        print("bank8 parity %d: Z by 8-point transforms vs per-channel sums, max |diff|" % par, bank8_parity_z(x, h, par))


# ---------------------------------------------------------------------------------------------------------------
# decimate-by-2 variant (round 3): fold the 2 aliases (k2 = k2' + 8 j, k2' in 0..7) and run a 2048-point inverse
#   k' = k0 + 16*k1 + 256*k2'                              m' = mu2 + 8*mu1 + 128*mu0    (mu2 in 0..7)
#   A: 8-point iDFT over k2' -> slot (i, mu2); twiddle conj W2048^((16*k1 + k0)*mu2)
#   two calls of the common tail, hb = mu2 >> 2, low = mu2 & 3:  lane = 4 mu1 + low, slot mu0: y_D[128 mu0 + 8 mu1 + 4 hb + low]
def inverse_dec2(p3):
    lane = np.arange(64)
    g, m = lane // 16, lane % 16
    out = np.zeros(2048, dtype=np.complex128)
    a8 = np.zeros((32, 64), dtype=np.complex128)         # slot (i, mu2) := 8 i + mu2
    for i in range(4):
        z = np.stack([p3[16 * i + k2p] + p3[16 * i + k2p + 8] for k2p in range(8)])
        seq = np.fft.ifft(z, axis=0) * 8
        for mu2 in range(8):
            a8[8 * i + mu2] = seq[mu2] * np.conj(W(2048, (16 * m + 4 * g + i) * mu2))
    for hb in range(2):
        a = np.zeros((16, 64), dtype=np.complex128)       # slot 4 i + low
        for i in range(4):
            for low in range(4):
                a[4 * i + low] = a8[8 * i + 4 * hb + low]
        xx = np.zeros_like(a)
        for gg in range(4):
            xx[:, 16 * gg:16 * gg + 16] = a[:, 16 * gg:16 * gg + 16].T
        j = lane % 16
        i_l, low_l = j // 4, j % 4
        k0_l = 4 * g + i_l
        bb = np.fft.ifft(xx, axis=0) * 16
        for mu1 in range(16):
            bb[mu1] = bb[mu1] * np.conj(W(256, k0_l * mu1))
        y = np.zeros_like(bb)
        for src in range(64):
            for mu1 in range(16):
                y[k0_l[src], 4 * mu1 + low_l[src]] = bb[mu1, src]
        c = np.fft.ifft(y, axis=0) * 16
        for mu0 in range(16):
            for ln in range(64):
                out[128 * mu0 + 8 * (ln >> 2) + 4 * hb + (ln & 3)] = c[mu0, ln]
    return out / N


def inverse_dec2_even_odd(p3):
    """the form the kernel uses: y[2p] = IFFT1024(Z[k] + Z[k+1024]), y[2p+1] = IFFT1024((Z[k] - Z[k+1024]) conj W2048^k), both
    through inverse_dec4's machinery (here: its input convention is slot (i, k2') with the 4 aliases ALREADY to be folded, so
    the even / odd spectra are placed at k2' = 0..3 and the other 12 alias slots are zero)"""
    lane = np.arange(64)
    g, m = lane // 16, lane % 16
    out = np.zeros(2048, dtype=np.complex128)
    ze = np.zeros_like(p3)
    zo = np.zeros_like(p3)
    for i in range(4):
        for q in range(4):
            u = p3[16 * i + q] + p3[16 * i + q + 8]          # fold the 2 aliases: Z(k2' = q)
            v = p3[16 * i + q + 4] + p3[16 * i + q + 12]     # Z(k2' = q + 4)
            k = (4 * g + i) + 16 * m + 256 * q
            ze[16 * i + q] = u + v
            zo[16 * i + q] = (u - v) * np.conj(W(2048, k))
    out[0::2] = inverse_dec4(ze)
    out[1::2] = inverse_dec4(zo)
    return out


def main_dec2():
    rng = np.random.default_rng(5)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    h = rng.standard_normal(255)
    H = np.fft.fft(h, N)
    Hp = np.zeros((64, 64), dtype=np.complex128)
    for slot in range(64):
        Hp[slot] = H[k_of(slot, np.arange(64))]
    yd = inverse_dec2(forward(x) * Hp)
    full = np.convolve(x, h)[:N]
    print("dec2 valid-part err", np.max(np.abs(yd[128:] - full[256::2])))
    ye = inverse_dec2_even_odd(forward(x) * Hp)
    print("dec2 (even/odd form) valid-part err", np.max(np.abs(ye[128:] - full[256::2])))


if __name__ == "__main__":
    main_dec2()


# ---------------------------------------------------------------------------------------------------------------
# odd decimation F = 3 (round 4, fir_odd_kernel): a block is F x 1024 input samples; lane l of row r loads the F consecutive samples
# x[s0 + F (64 r + l) + p]: F phase streams x_p[m] = x[s0 + F m + p] in the load layout of a 1024-point transform, and
#     y[F m] = sum_p (g_p * x_p)[m],   g_0[k] = h[F k],  g_p[d] = h[F d - p]  (p >= 1, d >= 1)
# is evaluated as  Z = sum_p FFT1024(x_p) G_p,  y = IFFT1024(Z).  forward_1024 is the mirror image of inverse_dec4 with the twiddles
# on the INPUTS of its passes (the kernel applies them in (cos, tan) form, fft16_tw):
#   A: FFT16 over the rows mu0 (m = 64 mu0 + 4 mu1 + mu2, lane = 4 mu1 + mu2) -> slot k0                         (plain)
#   Y^-1: element k0 of lane 4 mu1 + mu2 -> lane (k0, mu2) = 16 (k0 // 4) + 4 (k0 % 4) + mu2, slot mu1
#   B: FFT16 over mu1 -> k1, inputs carry (W256^k0)^mu1;   X^-1: element k1 of lane (g, j = 4 i + mu2) -> lane (g, k1), slot j
#   C: 4-point DFT over mu2 -> k2', inputs carry (W1024^(k0 + 16 k1))^mu2:  slot 4 i + k2' of lane (g, k1) = X[k0 + 16 k1 + 256 k2']
def forward_1024(v):
    lane = np.arange(64)
    g, m = lane // 16, lane % 16
    a = np.fft.fft(v, axis=0)
    y = np.zeros_like(a)
    for ln in range(64):
        mu1, mu2 = ln // 4, ln % 4
        for k0 in range(16):
            y[mu1, 16 * (k0 // 4) + 4 * (k0 % 4) + mu2] = a[k0, ln]
    j = lane % 16
    k0_l = 4 * g + j // 4
    b = np.fft.fft(y * (W(256, k0_l)[None, :] ** np.arange(16)[:, None]), axis=0)
    x = np.zeros_like(b)
    for gg in range(4):
        x[:, 16 * gg:16 * gg + 16] = b[:, 16 * gg:16 * gg + 16].T
    z = np.zeros_like(x)
    for i in range(4):
        base = W(1024, (4 * g + i) + 16 * m)
        z[4 * i:4 * i + 4] = np.fft.fft(x[4 * i:4 * i + 4] * base[None, :] ** np.arange(4)[:, None], axis=0)
    return z


def k1024_of(slot, lane):
    return (4 * (lane // 16) + slot // 4) + 16 * (lane % 16) + 256 * (slot % 4)


def odd_phase_taps(h, F, p):
    """g_p: the polyphase component of h that multiplies the phase stream x_p[m] = x[F m + p] (index = delay in outputs)"""
    gp = np.zeros(1024, dtype=np.asarray(h).dtype)
    for k in range(len(h)):
        if (k + p) % F == 0:
            gp[(k + p) // F] += h[k]
    return gp


def inverse_1024_from_z(z):
    """inverse_dec4 without its alias fold: z = slot (4 i + k2', lane (g, k1)) -> y[64 mu0 + lane]"""
    lane = np.arange(64)
    g = lane // 16
    a = np.zeros_like(z)
    for i in range(4):
        a[4 * i:4 * i + 4] = np.fft.ifft(z[4 * i:4 * i + 4], axis=0) * 4
    x = np.zeros_like(a)
    for gg in range(4):
        x[:, 16 * gg:16 * gg + 16] = a[:, 16 * gg:16 * gg + 16].T
    j = lane % 16
    k0_l, mu2_l = 4 * g + j // 4, j % 4
    b = np.fft.ifft(x * (np.conj(W(64, mu2_l))[None, :] ** np.arange(16)[:, None]), axis=0) * 16
    y = np.zeros_like(b)
    for src in range(64):
        for mu1 in range(16):
            y[k0_l[src], 4 * mu1 + mu2_l[src]] = b[mu1, src]
    c = np.fft.ifft(y * (np.conj(W(1024, lane))[None, :] ** np.arange(16)[:, None]), axis=0) * 16
    out = np.zeros(1024, dtype=np.complex128)
    for mu0 in range(16):
        out[64 * mu0 + lane] = c[mu0]
    return out / 1024


def odd_block(xblk, h, F):
    """y[F m], m = 0..1023, of one block of F x 1024 samples (valid from m = ceil((T - 1 + F - 1) / F) on)"""
    lane = np.arange(64)
    Z = 0
    for p in range(F):
        xp = xblk[p::F]
        Gp = np.fft.fft(odd_phase_taps(h, F, p))
        reg = np.stack([xp[64 * r + lane] for r in range(16)])
        Gl = np.stack([Gp[k1024_of(slot, lane)] for slot in range(16)])
        Z = Z + forward_1024(reg) * Gl
    return inverse_1024_from_z(Z)


def main_odd():
    rng = np.random.default_rng(6)
    for F in (3, 5):
        h = rng.standard_normal(255)
        xb = rng.standard_normal(F * 1024) + 1j * rng.standard_normal(F * 1024)
        y = odd_block(xb, h, F)
        full = np.convolve(xb, h)[:F * 1024][::F]
        ovl = -(-(255 - 1 + F - 1) // F)
        print("odd block F = %d: valid-part err %.3g" % (F, np.max(np.abs(y[ovl:] - full[ovl:]))))


if __name__ == "__main__":
    main_odd()
