// jpeg_decode.cpp -- see jpeg_decode.h.  Own implementation of ITU-T T.81 Huffman decoding
// (sequential and progressive) with libjpeg-compatible reconstruction arithmetic.
#include "jpeg_decode.h"

namespace mpmvs_host {
int OmpThreads();  // planar_prior.cpp: min(16, hardware threads), never the OpenMP default (PatchMatch.h)
}

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <new>

namespace {

const uint8_t kZigzag[64 + 16] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                                  6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                                  39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                                  // a corrupt run may step past 63; those land on the last coefficient instead of outside the block
                                  63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct Huff {
    bool defined = false;
    uint8_t bits[17] = {0};
    uint8_t vals[256] = {0};
    uint16_t fast[512];
    int32_t maxcode[18];
    int32_t valoff[17];
    void build() {
        std::memset(fast, 0, sizeof(fast));
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            valoff[l] = k - code;
            for (int i = 0; i < bits[l]; ++i, ++code, ++k) {
                if (l <= 9) {
                    const int first = code << (9 - l);
                    for (int f = 0; f < (1 << (9 - l)); ++f) fast[first + f] = (uint16_t)((l << 8) | vals[k]);
                }
            }
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7FFFFFFF;
        defined = true;
    }
};

// entropy-coded segment reader: bytes are unstuffed (FF 00 -> FF); at a marker the stream
// yields zero bits, as libjpeg does for truncated data
struct BitReader {
    const uint8_t* p = nullptr;
    const uint8_t* end = nullptr;
    uint64_t acc = 0;
    int nbits = 0;
    bool at_marker = false;
    void reset(const uint8_t* at) {
        p = at;
        acc = 0;
        nbits = 0;
        at_marker = false;
    }
    void fill() {
        while (nbits <= 56) {
            unsigned b = 0;
            if (!at_marker) {
                if (p >= end) {
                    at_marker = true;
                } else if (*p != 0xFF) {
                    b = *p++;
                } else if (p + 1 < end && p[1] == 0x00) {
                    b = 0xFF;
                    p += 2;
                } else {
                    at_marker = true;
                }
            }
            acc |= (uint64_t)b << (56 - nbits);
            nbits += 8;
        }
    }
    inline unsigned peek(int n) {
        if (nbits < n) fill();
        return (unsigned)(acc >> (64 - n));
    }
    inline void skip(int n) {
        acc <<= n;
        nbits -= n;
    }
    inline unsigned get(int n) {
        if (n == 0) return 0;
        const unsigned v = peek(n);
        skip(n);
        return v;
    }
    inline int decode(const Huff& h) {
        if (nbits < 16) fill();
        const unsigned e = h.fast[acc >> 55];
        if (e) {
            skip(e >> 8);
            return e & 255;
        }
        const unsigned look = (unsigned)(acc >> 48);
        int l = 10;
        while (l <= 16 && (int)(look >> (16 - l)) > h.maxcode[l]) ++l;
        if (l > 16) {  // not a code of this table (corrupt data): libjpeg substitutes 0
            skip(16);
            return 0;
        }
        skip(l);
        return h.vals[((look >> (16 - l)) + h.valoff[l]) & 255];
    }
};

inline int extend(int r, int s) { return r < (1 << (s - 1)) ? r - (1 << s) + 1 : r; }

struct Comp {
    int id = 0, h = 1, v = 1, tq = 0;
    int wb = 0, hb = 0;  // blocks, padded to whole MCUs
    int cw = 0, ch = 0;  // real (downsampled) size in samples
    int td = 0, ta = 0, dc_pred = 0;
    bool q_latched = false;
    uint16_t q[64];
    std::vector<int16_t> coef;
    std::vector<uint8_t> plane;  // wb*8 x hb*8 after the inverse DCT
};

struct Decoder {
    const uint8_t* data;
    size_t size;
    std::string err;
    uint16_t qt[4][64];
    bool qt_defined[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    int W = 0, H = 0, ncomp = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
    bool progressive = false, have_frame = false;
    int restart_interval = 0;
    int adobe_transform = -1;
    int orientation = 1;  // EXIF tag 0x0112 (cv::imread applies it unless IMREAD_IGNORE_ORIENTATION is given)
    Comp comp[3];

    bool fail(const char* m) {
        if (err.empty()) err = m;
        return false;
    }

    bool parse();
    bool read_dqt(const uint8_t* s, int len);
    bool read_dht(const uint8_t* s, int len);
    bool read_sof(const uint8_t* s, int len);
    bool read_scan(const uint8_t* s, int len, size_t& pos);
    void read_exif(const uint8_t* s, int len);
    void reconstruct();
};

bool Decoder::read_dqt(const uint8_t* s, int len) {
    int i = 0;
    while (i < len) {
        const int pq = s[i] >> 4, tq = s[i] & 15;
        ++i;
        if (tq > 3 || pq > 1) return fail("bad quantisation table header");
        if (i + 64 * (pq + 1) > len) return fail("truncated quantisation table");
        for (int k = 0; k < 64; ++k) {
            const int val = pq ? (s[i] << 8 | s[i + 1]) : s[i];
            i += pq + 1;
            qt[tq][kZigzag[k]] = (uint16_t)val;
        }
        qt_defined[tq] = true;
    }
    return true;
}

bool Decoder::read_dht(const uint8_t* s, int len) {
    int i = 0;
    while (i < len) {
        if (i + 17 > len) return fail("truncated Huffman table");
        const int tc = s[i] >> 4, th = s[i] & 15;
        if (tc > 1 || th > 3) return fail("bad Huffman table header");
        Huff& h = tc ? ac[th] : dc[th];
        int n = 0;
        h.bits[0] = 0;
        for (int l = 1; l <= 16; ++l) {
            h.bits[l] = s[i + l];
            n += h.bits[l];
        }
        i += 17;
        if (n > 256 || i + n > len) return fail("bad Huffman table size");
        std::memset(h.vals, 0, sizeof(h.vals));
        std::memcpy(h.vals, s + i, n);
        i += n;
        h.build();
    }
    return true;
}

bool Decoder::read_sof(const uint8_t* s, int len) {
    if (have_frame) return fail("more than one frame header");
    if (len < 6) return fail("truncated frame header");
    if (s[0] != 8) return fail("only 8-bit JPEG files are supported");
    H = s[1] << 8 | s[2];
    W = s[3] << 8 | s[4];
    ncomp = s[5];
    if (W <= 0 || H <= 0) return fail("empty image (or DNL-defined height, unsupported)");
    if ((int64_t)W * H > ((int64_t)1 << 28)) return fail("image larger than 2^28 pixels");
    if (ncomp != 1 && ncomp != 3) return fail("only 1- and 3-component JPEG files are supported");
    if (len < 6 + 3 * ncomp) return fail("truncated frame header");
    hmax = vmax = 1;
    for (int c = 0; c < ncomp; ++c) {
        Comp& k = comp[c];
        k.id = s[6 + 3 * c];
        k.h = s[7 + 3 * c] >> 4;
        k.v = s[7 + 3 * c] & 15;
        k.tq = s[8 + 3 * c];
        if (k.h < 1 || k.h > 4 || k.v < 1 || k.v > 4 || k.tq > 3) return fail("bad component parameters");
        hmax = std::max(hmax, k.h);
        vmax = std::max(vmax, k.v);
    }
    if (ncomp == 1) comp[0].h = comp[0].v = hmax = vmax = 1;  // a single component is never interleaved (T.81 A.2.2)
    mcux = (W + 8 * hmax - 1) / (8 * hmax);
    mcuy = (H + 8 * vmax - 1) / (8 * vmax);
    for (int c = 0; c < ncomp; ++c) {
        Comp& k = comp[c];
        if (hmax % k.h || vmax % k.v) return fail("fractional sampling ratios are not supported");
        k.wb = mcux * k.h;
        k.hb = mcuy * k.v;
        k.cw = (W * k.h + hmax - 1) / hmax;
        k.ch = (H * k.v + vmax - 1) / vmax;
        k.coef.assign((size_t)k.wb * k.hb * 64, 0);
    }
    have_frame = true;
    return true;
}

// APP1 "Exif\0\0" + TIFF header + IFD0: only the orientation tag is of interest
void Decoder::read_exif(const uint8_t* s, int len) {
    if (len < 14 || std::memcmp(s, "Exif\0\0", 6) != 0) return;
    const uint8_t* t = s + 6;
    const int n = len - 6;
    const bool le = t[0] == 'I' && t[1] == 'I';
    if (!le && !(t[0] == 'M' && t[1] == 'M')) return;
    auto u16 = [&](int o) { return le ? (t[o] | t[o + 1] << 8) : (t[o] << 8 | t[o + 1]); };
    auto u32 = [&](int o) { return le ? ((uint32_t)t[o] | (uint32_t)t[o + 1] << 8 | (uint32_t)t[o + 2] << 16 | (uint32_t)t[o + 3] << 24)
                                      : ((uint32_t)t[o] << 24 | (uint32_t)t[o + 1] << 16 | (uint32_t)t[o + 2] << 8 | (uint32_t)t[o + 3]); };
    if (u16(2) != 42) return;
    const uint32_t ifd = u32(4);
    if (ifd + 2 > (uint32_t)n) return;
    const int entries = u16((int)ifd);
    for (int e = 0; e < entries; ++e) {
        const uint32_t o = ifd + 2 + 12u * e;
        if (o + 12 > (uint32_t)n) return;
        if (u16((int)o) == 0x0112 && u16((int)o + 2) == 3) {  // SHORT
            const int v = u16((int)o + 8);
            if (v >= 1 && v <= 8) orientation = v;
            return;
        }
    }
}

// ---- one scan ----------------------------------------------------------------------------
bool Decoder::read_scan(const uint8_t* s, int len, size_t& pos) {
    if (!have_frame) return fail("scan before frame header");
    if (len < 1) return fail("truncated scan header");
    const int ns = s[0];
    if (ns < 1 || ns > ncomp || len < 4 + 2 * ns) return fail("bad scan header");
    Comp* sc[3];
    for (int i = 0; i < ns; ++i) {
        sc[i] = nullptr;
        for (int c = 0; c < ncomp; ++c)
            if (comp[c].id == s[1 + 2 * i]) sc[i] = &comp[c];
        if (!sc[i]) return fail("scan names an unknown component");
        sc[i]->td = s[2 + 2 * i] >> 4;
        sc[i]->ta = s[2 + 2 * i] & 15;
        if (sc[i]->td > 3 || sc[i]->ta > 3) return fail("bad table selector");
        if (!sc[i]->q_latched) {
            if (!qt_defined[sc[i]->tq]) return fail("quantisation table missing");
            std::memcpy(sc[i]->q, qt[sc[i]->tq], sizeof(sc[i]->q));
            sc[i]->q_latched = true;
        }
    }
    int Ss = s[1 + 2 * ns], Se = s[2 + 2 * ns];
    const int Ah = s[3 + 2 * ns] >> 4, Al = s[3 + 2 * ns] & 15;
    if (progressive) {
        if (Ss > Se || Se > 63 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1) || Al > 13) return fail("bad progressive scan parameters");
    } else {
        Ss = 0;
        Se = 63;
    }
    for (int i = 0; i < ns; ++i) {
        const bool need_dc = !progressive || (Ss == 0 && Ah == 0);
        const bool need_ac = !progressive || Ss > 0;
        if (need_dc && !dc[sc[i]->td].defined) return fail("DC Huffman table missing");
        if (need_ac && !ac[sc[i]->ta].defined) return fail("AC Huffman table missing");
    }

    BitReader br;
    br.end = data + size;
    br.reset(data + pos);
    int mx, my;
    if (ns == 1) {
        mx = (sc[0]->cw + 7) / 8;
        my = (sc[0]->ch + 7) / 8;
    } else {
        mx = mcux;
        my = mcuy;
    }
    int eobrun = 0, to_restart = restart_interval, next_rst = 0;
    for (int i = 0; i < ns; ++i) sc[i]->dc_pred = 0;
    const int p1 = 1 << Al, m1 = -(1 << Al);

    for (int row = 0; row < my; ++row) {
        for (int col = 0; col < mx; ++col) {
            if (restart_interval && to_restart == 0) {
                // resynchronise on the next RSTn marker
                const uint8_t* q = br.p;
                while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) {
                    if (q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF) break;  // some other marker: give up resync
                    ++q;
                }
                if (q + 1 < br.end && q[0] == 0xFF && q[1] == 0xD0 + next_rst) q += 2;
                next_rst = (next_rst + 1) & 7;
                br.reset(q);
                for (int i = 0; i < ns; ++i) sc[i]->dc_pred = 0;
                eobrun = 0;
                to_restart = restart_interval;
            }
            for (int i = 0; i < ns; ++i) {
                Comp& k = *sc[i];
                const int bw = ns == 1 ? 1 : k.h, bh = ns == 1 ? 1 : k.v;
                for (int by = 0; by < bh; ++by)
                    for (int bx = 0; bx < bw; ++bx) {
                        int16_t* blk = &k.coef[((size_t)(row * bh + by) * k.wb + (col * bw + bx)) * 64];
                        if (!progressive) {
                            int t = br.decode(dc[k.td]) & 15;
                            int diff = t ? extend((int)br.get(t), t) : 0;
                            k.dc_pred += diff;
                            blk[0] = (int16_t)k.dc_pred;
                            for (int kk = 1; kk < 64;) {
                                const int rs = br.decode(ac[k.ta]);
                                const int r = rs >> 4, sz = rs & 15;
                                if (sz) {
                                    kk += r;
                                    blk[kZigzag[kk]] = (int16_t)extend((int)br.get(sz), sz);
                                    ++kk;
                                } else {
                                    if (r != 15) break;
                                    kk += 16;
                                }
                            }
                        } else if (Ss == 0) {
                            if (Ah == 0) {
                                int t = br.decode(dc[k.td]) & 15;
                                int diff = t ? extend((int)br.get(t), t) : 0;
                                k.dc_pred += diff;
                                blk[0] = (int16_t)(k.dc_pred * (1 << Al));
                            } else if (br.get(1)) {
                                blk[0] |= (int16_t)p1;
                            }
                        } else if (Ah == 0) {
                            // AC first pass (T.81 G.1.2.2)
                            if (eobrun > 0) {
                                --eobrun;
                            } else {
                                for (int kk = Ss; kk <= Se; ++kk) {
                                    const int rs = br.decode(ac[k.ta]);
                                    const int r = rs >> 4, sz = rs & 15;
                                    if (sz) {
                                        kk += r;
                                        blk[kZigzag[kk]] = (int16_t)(extend((int)br.get(sz), sz) * (1 << Al));
                                    } else if (r == 15) {
                                        kk += 15;
                                    } else {
                                        eobrun = 1 << r;
                                        if (r) eobrun += (int)br.get(r);
                                        --eobrun;
                                        break;
                                    }
                                }
                            }
                        } else {
                            // AC refinement pass (T.81 G.1.2.3)
                            int kk = Ss;
                            if (eobrun == 0) {
                                for (; kk <= Se; ++kk) {
                                    const int rs = br.decode(ac[k.ta]);
                                    int r = rs >> 4, sz = rs & 15;
                                    int val = 0;
                                    if (sz) {
                                        val = br.get(1) ? p1 : m1;
                                    } else if (r != 15) {
                                        eobrun = 1 << r;
                                        if (r) eobrun += (int)br.get(r);
                                        break;
                                    }
                                    // skip r zero-history coefficients, refining the non-zero ones passed on the way
                                    do {
                                        int16_t* c = blk + kZigzag[kk];
                                        if (*c != 0) {
                                            if (br.get(1) && (*c & p1) == 0) *c = (int16_t)(*c + (*c >= 0 ? p1 : m1));
                                        } else if (--r < 0) {
                                            break;
                                        }
                                        ++kk;
                                    } while (kk <= Se);
                                    if (val && kk <= 63) blk[kZigzag[kk]] = (int16_t)val;
                                }
                            }
                            if (eobrun > 0) {
                                for (; kk <= Se; ++kk) {
                                    int16_t* c = blk + kZigzag[kk];
                                    if (*c != 0 && br.get(1) && (*c & p1) == 0) *c = (int16_t)(*c + (*c >= 0 ? p1 : m1));
                                }
                                --eobrun;
                            }
                        }
                    }
            }
            if (restart_interval) --to_restart;
        }
    }
    // continue marker parsing after the entropy-coded segment
    const uint8_t* q = br.p;
    while (q + 1 < br.end && !(q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF && !(q[1] >= 0xD0 && q[1] <= 0xD7))) ++q;
    pos = (size_t)(q - data);
    return true;
}

// ---- reconstruction -----------------------------------------------------------------------
// libjpeg's range-limit table, indexed modulo 1024 about the +128 level shift
inline uint8_t range_limit(int64_t x) {
    const int i = (int)(x & 1023);
    if (i < 128) return (uint8_t)(i + 128);
    if (i < 512) return 255;
    if (i < 896) return 0;
    return (uint8_t)(i - 896);
}

// accurate integer inverse DCT (Loeffler-Ligtenberg-Moschytz, 13-bit constants, 2 extra bits
// kept between the passes) -- the algorithm of libjpeg's default JDCT_ISLOW
void idct_islow(const int16_t* in, const uint16_t* q, uint8_t* out, int stride) {
    constexpr int CB = 13, P1 = 2;
    constexpr int64_t F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299, F1_847 = 15137,
                      F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
    int64_t ws[64];
    auto descale = [](int64_t x, int n) { return (x + ((int64_t)1 << (n - 1))) >> n; };
    for (int c = 0; c < 8; ++c) {
        auto d = [&](int r) { return (int64_t)in[r * 8 + c] * q[r * 8 + c]; };
        int64_t z2 = d(2), z3 = d(6);
        int64_t z1 = (z2 + z3) * F0_541;
        int64_t t2 = z1 + z3 * (-F1_847), t3 = z1 + z2 * F0_765;
        z2 = d(0);
        z3 = d(4);
        int64_t t0 = (z2 + z3) * ((int64_t)1 << CB), t1 = (z2 - z3) * ((int64_t)1 << CB);
        const int64_t t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
        t0 = d(7);
        t1 = d(5);
        t2 = d(3);
        t3 = d(1);
        z1 = t0 + t3;
        z2 = t1 + t2;
        z3 = t0 + t2;
        int64_t z4 = t1 + t3;
        const int64_t z5 = (z3 + z4) * F1_175;
        t0 *= F0_298;
        t1 *= F2_053;
        t2 *= F3_072;
        t3 *= F1_501;
        z1 *= -F0_899;
        z2 *= -F2_562;
        z3 *= -F1_961;
        z4 *= -F0_390;
        z3 += z5;
        z4 += z5;
        t0 += z1 + z3;
        t1 += z2 + z4;
        t2 += z2 + z3;
        t3 += z1 + z4;
        ws[0 * 8 + c] = descale(t10 + t3, CB - P1);
        ws[7 * 8 + c] = descale(t10 - t3, CB - P1);
        ws[1 * 8 + c] = descale(t11 + t2, CB - P1);
        ws[6 * 8 + c] = descale(t11 - t2, CB - P1);
        ws[2 * 8 + c] = descale(t12 + t1, CB - P1);
        ws[5 * 8 + c] = descale(t12 - t1, CB - P1);
        ws[3 * 8 + c] = descale(t13 + t0, CB - P1);
        ws[4 * 8 + c] = descale(t13 - t0, CB - P1);
    }
    for (int r = 0; r < 8; ++r) {
        const int64_t* w = ws + r * 8;
        int64_t z2 = w[2], z3 = w[6];
        int64_t z1 = (z2 + z3) * F0_541;
        int64_t t2 = z1 + z3 * (-F1_847), t3 = z1 + z2 * F0_765;
        int64_t t0 = (w[0] + w[4]) * ((int64_t)1 << CB), t1 = (w[0] - w[4]) * ((int64_t)1 << CB);
        const int64_t t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
        t0 = w[7];
        t1 = w[5];
        t2 = w[3];
        t3 = w[1];
        z1 = t0 + t3;
        z2 = t1 + t2;
        z3 = t0 + t2;
        int64_t z4 = t1 + t3;
        const int64_t z5 = (z3 + z4) * F1_175;
        t0 *= F0_298;
        t1 *= F2_053;
        t2 *= F3_072;
        t3 *= F1_501;
        z1 *= -F0_899;
        z2 *= -F2_562;
        z3 *= -F1_961;
        z4 *= -F0_390;
        z3 += z5;
        z4 += z5;
        t0 += z1 + z3;
        t1 += z2 + z4;
        t2 += z2 + z3;
        t3 += z1 + z4;
        constexpr int S = CB + P1 + 3;
        uint8_t* o = out + (size_t)r * stride;
        o[0] = range_limit(descale(t10 + t3, S));
        o[7] = range_limit(descale(t10 - t3, S));
        o[1] = range_limit(descale(t11 + t2, S));
        o[6] = range_limit(descale(t11 - t2, S));
        o[2] = range_limit(descale(t12 + t1, S));
        o[5] = range_limit(descale(t12 - t1, S));
        o[3] = range_limit(descale(t13 + t0, S));
        o[4] = range_limit(descale(t13 - t0, S));
    }
}

void Decoder::reconstruct() {
    for (int c = 0; c < ncomp; ++c) {
        Comp& k = comp[c];
        const int stride = k.wb * 8;
        k.plane.assign((size_t)stride * k.hb * 8, 0);
        if (!k.q_latched) {
            std::memset(k.q, 0, sizeof(k.q));
        }
#pragma omp parallel for num_threads(mpmvs_host::OmpThreads()) schedule(static)
        for (int by = 0; by < k.hb; ++by)
            for (int bx = 0; bx < k.wb; ++bx) idct_islow(&k.coef[((size_t)by * k.wb + bx) * 64], k.q, &k.plane[(size_t)by * 8 * stride + bx * 8], stride);
    }
}

// chroma upsampling to the full image grid; the 2:1 cases use libjpeg's "fancy" triangle
// filter (3/4 nearer + 1/4 further sample, alternating rounding bias), neighbours clamped to
// the component's real extent; other ratios replicate samples
void upsample(const Comp& k, int hmax, int vmax, int W, int H, std::vector<uint8_t>& out) {
    const int fh = hmax / k.h, fv = vmax / k.v;
    const int stride = k.wb * 8;
    out.resize((size_t)W * H);
    const uint8_t* in = k.plane.data();
    auto row = [&](int r) { return in + (size_t)std::min(std::max(r, 0), k.ch - 1) * stride; };
    auto col = [&](int c) { return std::min(std::max(c, 0), k.cw - 1); };
    if (fh == 1 && fv == 1) {
        for (int y = 0; y < H; ++y) std::memcpy(&out[(size_t)y * W], row(y), W);
    } else if (fh == 2 && fv == 1 && k.cw > 2) {
        for (int y = 0; y < H; ++y) {
            const uint8_t* s = row(y);
            uint8_t* o = &out[(size_t)y * W];
            for (int x = 0; x < W; ++x) {
                const int i = x >> 1;
                o[x] = (x & 1) ? (uint8_t)((3 * s[i] + s[col(i + 1)] + 2) >> 2) : (uint8_t)((3 * s[i] + s[col(i - 1)] + 1) >> 2);
            }
        }
    } else if (fh == 2 && fv == 2 && k.cw > 2) {
#pragma omp parallel for num_threads(mpmvs_host::OmpThreads()) schedule(static)
        for (int y = 0; y < H; ++y) {
            const int r = y >> 1;
            const uint8_t* s0 = row(r);
            const uint8_t* s1 = row((y & 1) ? r + 1 : r - 1);
            uint8_t* o = &out[(size_t)y * W];
            for (int x = 0; x < W; ++x) {
                const int i = x >> 1;
                const int cur = 3 * s0[i] + s1[i];
                if (x & 1) {
                    const int j = col(i + 1);
                    o[x] = (uint8_t)((3 * cur + 3 * s0[j] + s1[j] + 7) >> 4);
                } else {
                    const int j = col(i - 1);
                    o[x] = (uint8_t)((3 * cur + 3 * s0[j] + s1[j] + 8) >> 4);
                }
            }
        }
    } else if (fh == 1 && fv == 2) {
        for (int y = 0; y < H; ++y) {
            const int r = y >> 1;
            const uint8_t* s0 = row(r);
            const uint8_t* s1 = row((y & 1) ? r + 1 : r - 1);
            const int bias = (y & 1) ? 2 : 1;
            uint8_t* o = &out[(size_t)y * W];
            for (int x = 0; x < W; ++x) o[x] = (uint8_t)((3 * s0[x] + s1[x] + bias) >> 2);
        }
    } else {
        for (int y = 0; y < H; ++y) {
            const uint8_t* s = row(y / fv);
            uint8_t* o = &out[(size_t)y * W];
            for (int x = 0; x < W; ++x) o[x] = s[x / fh];
        }
    }
}

bool Decoder::parse() {
    if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return fail("not a JPEG file (no SOI marker)");
    size_t pos = 2;
    bool seen_scan = false;
    while (true) {
        while (pos < size && data[pos] != 0xFF) ++pos;  // tolerate garbage between segments
        while (pos < size && data[pos] == 0xFF) ++pos;
        if (pos >= size) break;
        const int m = data[pos++];
        if (m == 0xD9) break;                              // EOI
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;  // TEM, stray RSTn
        // a file cut short after its first scan still yields an image (libjpeg: "premature end of file" warning)
        if (pos + 2 > size) {
            if (seen_scan) break;
            return fail("truncated marker segment");
        }
        const int len = (data[pos] << 8 | data[pos + 1]) - 2;
        if (len < 0 || pos + 2 + (size_t)len > size) {
            if (seen_scan) break;
            return fail("truncated marker segment");
        }
        const uint8_t* s = data + pos + 2;
        pos += 2 + (size_t)len;
        switch (m) {
            case 0xDB:
                if (!read_dqt(s, len)) return false;
                break;
            case 0xC4:
                if (!read_dht(s, len)) return false;
                break;
            case 0xC0:
            case 0xC1:
                progressive = false;
                if (!read_sof(s, len)) return false;
                break;
            case 0xC2:
                progressive = true;
                if (!read_sof(s, len)) return false;
                break;
            case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                return fail("unsupported JPEG process (lossless, hierarchical or arithmetic coding)");
            case 0xDD:
                if (len < 2) return fail("truncated restart interval");
                restart_interval = s[0] << 8 | s[1];
                break;
            case 0xDA:
                if (!read_scan(s, len, pos)) return false;
                seen_scan = true;
                break;
            case 0xEE:
                if (len >= 12 && std::memcmp(s, "Adobe", 5) == 0) adobe_transform = s[11];
                break;
            case 0xE1:
                read_exif(s, len);
                break;
            default:
                break;  // APPn, COM, DNL, ...
        }
    }
    if (!have_frame || !seen_scan) return fail("no image data");
    return true;
}

}  // namespace

// EXIF orientations 2..8 -> upright image (what cv::imread returns): 2 mirror, 3 rotate 180, 4 flip, 5 transpose,
// 6 rotate 90 clockwise, 7 transverse, 8 rotate 90 counter-clockwise
static void apply_orientation(int orientation, int channels, std::vector<uint8_t>& px, int& w, int& h) {
    if (orientation <= 1 || orientation > 8) return;
    const bool swap = orientation >= 5;
    const int nw = swap ? h : w, nh = swap ? w : h;
    std::vector<uint8_t> out(px.size());
    for (int y = 0; y < nh; ++y)
        for (int x = 0; x < nw; ++x) {
            int sx, sy;
            switch (orientation) {
                case 2: sx = w - 1 - x; sy = y; break;
                case 3: sx = w - 1 - x; sy = h - 1 - y; break;
                case 4: sx = x; sy = h - 1 - y; break;
                case 5: sx = y; sy = x; break;
                case 6: sx = y; sy = h - 1 - x; break;
                case 7: sx = w - 1 - y; sy = h - 1 - x; break;
                default: sx = w - 1 - y; sy = x; break;  // 8
            }
            std::memcpy(&out[((size_t)y * nw + x) * channels], &px[((size_t)sy * w + sx) * channels], channels);
        }
    px.swap(out);
    w = nw;
    h = nh;
}

bool DecodeJpeg(const uint8_t* data, size_t size, int channels, std::vector<uint8_t>& pixels, int& width, int& height, std::string& err) {
    if (channels != 1 && channels != 3) {
        err = "channels must be 1 or 3";
        return false;
    }
    Decoder d;
    d.data = data;
    d.size = size;
    try {
        if (!d.parse()) {
            err = d.err;
            return false;
        }
    } catch (const std::bad_alloc&) {
        err = "out of memory";
        return false;
    }
    if (d.ncomp == 3) {
        const bool rgb_ids = d.comp[0].id == 'R' && d.comp[1].id == 'G' && d.comp[2].id == 'B';
        if (d.adobe_transform == 0 || (d.adobe_transform < 0 && rgb_ids)) {
            err = "RGB-coded JPEG files (no YCbCr transform) are not supported";
            return false;
        }
    }
    d.reconstruct();
    width = d.W;
    height = d.H;
    const int W = d.W, H = d.H;
    std::vector<uint8_t> y, cb, cr;
    upsample(d.comp[0], d.hmax, d.vmax, W, H, y);
    if (channels == 1) {
        pixels.swap(y);
        apply_orientation(d.orientation, 1, pixels, width, height);
        return true;
    }
    pixels.resize((size_t)W * H * 3);
    if (d.ncomp == 1) {
        for (size_t i = 0; i < (size_t)W * H; ++i) pixels[3 * i] = pixels[3 * i + 1] = pixels[3 * i + 2] = y[i];
        apply_orientation(d.orientation, 3, pixels, width, height);
        return true;
    }
    upsample(d.comp[1], d.hmax, d.vmax, W, H, cb);
    upsample(d.comp[2], d.hmax, d.vmax, W, H, cr);
    // YCbCr -> RGB with libjpeg's 16-bit fixed-point tables
    int cr_r[256], cb_b[256];
    int64_t cr_g[256], cb_g[256];
    auto fix = [](double x) { return (int64_t)(x * 65536.0 + 0.5); };
    for (int i = 0; i < 256; ++i) {
        const int64_t x = i - 128;
        cr_r[i] = (int)((fix(1.40200) * x + 32768) >> 16);
        cb_b[i] = (int)((fix(1.77200) * x + 32768) >> 16);
        cr_g[i] = -fix(0.71414) * x;
        cb_g[i] = -fix(0.34414) * x + 32768;
    }
    auto clamp8 = [](int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); };
#pragma omp parallel for num_threads(mpmvs_host::OmpThreads()) schedule(static)
    for (int r = 0; r < H; ++r) {
        for (int c = 0; c < W; ++c) {
            const size_t i = (size_t)r * W + c;
            const int yy = y[i];
            pixels[3 * i + 2] = clamp8(yy + cr_r[cr[i]]);
            pixels[3 * i + 1] = clamp8(yy + (int)((cb_g[cb[i]] + cr_g[cr[i]]) >> 16));
            pixels[3 * i + 0] = clamp8(yy + cb_b[cb[i]]);
        }
    }
    apply_orientation(d.orientation, 3, pixels, width, height);
    return true;
}

bool DecodeJpegFile(const std::string& path, int channels, std::vector<uint8_t>& pixels, int& width, int& height, std::string& err) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) {
        err = "cannot open " + path;
        return false;
    }
    std::vector<uint8_t> buf;
    uint8_t chunk[1 << 16];
    size_t n;
    while ((n = std::fread(chunk, 1, sizeof(chunk), f)) > 0) buf.insert(buf.end(), chunk, chunk + n);
    std::fclose(f);
    return DecodeJpeg(buf.data(), buf.size(), channels, pixels, width, height, err);
}
