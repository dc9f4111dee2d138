// Baseline (Huffman, sequential DCT, 8-bit) JPEG -> grey: see jpeg.h.  Written from the standard (ITU-T T.81: marker syntax,
// Huffman procedures of Annex F, table construction of Annex C) and the published description of libjpeg's "islow" inverse DCT
// (Loeffler-Ligtenberg-Moschytz, 13-bit constants, two passes with 2 extra bits after the first) so that the samples are libjpeg's.
#include "jpeg.h"

#include <algorithm>
#include <cstring>
#include <vector>

namespace LpSlam {
namespace {

const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTable {
    bool present = false;
    uint8_t bits[17] = {0};          // codes of each length 1..16
    uint8_t vals[256] = {0};
    // decoding (T.81 F.2.2.3): smallest / largest code of every length and the index of its first value
    int32_t mincode[17], maxcode[18], valptr[17];
    // 9-bit look-ahead: (length << 8) | symbol, 0 = longer than 9 bits
    uint16_t look[512];
    bool build()
    {
        int code = 0, k = 0;
        std::memset(look, 0, sizeof(look));
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k; mincode[l] = code;
            for (int i = 0; i < bits[l]; ++i, ++k, ++code) {
                if (k >= 256) return false;
                if (l <= 9) {
                    const int first = code << (9 - l), n = 1 << (9 - l);
                    if (first + n > 512) return false;
                    for (int j = 0; j < n; ++j) look[first + j] = (uint16_t)((l << 8) | vals[k]);
                }
            }
            maxcode[l] = bits[l] ? code - 1 : -1;
            if (code > (1 << l)) return false;                      // over-subscribed
            code <<= 1;
        }
        maxcode[17] = 0x7FFFFFFF;
        return true;
    }
};

struct BitReader {
    const uint8_t* p; const uint8_t* end;
    uint64_t acc = 0; int n = 0;         // n valid bits at the top of acc's low 64 - ... kept right-aligned
    int marker = 0;                      // a marker met inside the entropy-coded data (0 = none): zeros are fed from then on
    void fill()
    {
        while (n <= 48) {
            int b = 0;
            if (!marker && p < end) {
                b = *p++;
                if (b == 0xFF) {
                    int b2 = p < end ? *p : 0xD9;
                    while (b2 == 0xFF && p < end) { ++p; b2 = p < end ? *p : 0xD9; }       // fill bytes
                    if (b2 == 0) ++p;                                 // stuffed zero: a data byte FF
                    else { marker = b2; ++p; b = 0; }
                }
            }
            acc = (acc << 8) | (uint64_t)b; n += 8;
        }
    }
    int peek(int k) { if (n < k) fill(); return (int)((acc >> (n - k)) & ((1u << k) - 1)); }
    void skip(int k) { n -= k; }
    int get(int k) { if (k == 0) return 0; const int v = peek(k); skip(k); return v; }
    void restart() { acc = 0; n = 0; marker = 0; }
};

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

bool decode_symbol(BitReader& br, const HuffTable& t, int& sym)
{
    const int look = br.peek(9);
    const uint16_t e = t.look[look];
    if (e) { br.skip(e >> 8); sym = e & 255; return true; }
    int code = look, l = 9;
    while (l < 16) {
        ++l;
        code = br.peek(l);
        if (t.maxcode[l] >= 0 && code <= t.maxcode[l] && code >= t.mincode[l]) { br.skip(l); sym = t.vals[t.valptr[l] + code - t.mincode[l]]; return true; }
    }
    return false;
}

// libjpeg's jpeg_idct_islow: coefficients (natural order) x quantisation table -> 8x8 samples
constexpr int CONST_BITS = 13, PASS1_BITS = 2;
constexpr int32_t F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270, F_0_899976223 = 7373, F_1_175875602 = 9633,
                  F_1_501321110 = 12299, F_1_847759065 = 15137, F_1_961570560 = 16069, F_2_053119869 = 16819, F_2_562915447 = 20995, F_3_072711026 = 25172;
inline int32_t descale(int64_t x, int n) { return (int32_t)((x + ((int64_t)1 << (n - 1))) >> n); }
inline uint8_t range_limit(int32_t x)
{
    const int32_t xs = ((x & 1023) ^ 512) - 512;                      // the table is indexed modulo 1024
    const int32_t v = xs + 128;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

void idct_islow(const int16_t* coef, const uint16_t* quant, uint8_t* out, int pitch)
{
    int32_t ws[64];
    for (int c = 0; c < 8; ++c) {
        const int16_t* in = coef + c; const uint16_t* q = quant + c; int32_t* w = ws + c;
        if (in[8] == 0 && in[16] == 0 && in[24] == 0 && in[32] == 0 && in[40] == 0 && in[48] == 0 && in[56] == 0) {
            const int32_t dc = (int32_t)((uint32_t)((int32_t)in[0] * (int32_t)q[0]) << PASS1_BITS);
            for (int r = 0; r < 8; ++r) w[8 * r] = dc;
            continue;
        }
        int64_t z2 = (int32_t)in[16] * (int32_t)q[16], z3 = (int32_t)in[48] * (int32_t)q[48];
        int64_t z1 = (z2 + z3) * F_0_541196100;
        int64_t tmp2 = z1 + z3 * (-F_1_847759065), tmp3 = z1 + z2 * F_0_765366865;
        z2 = (int32_t)in[0] * (int32_t)q[0]; z3 = (int32_t)in[32] * (int32_t)q[32];
        int64_t tmp0 = (z2 + z3) * (1 << CONST_BITS), tmp1 = (z2 - z3) * (1 << CONST_BITS);
        const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = (int32_t)in[56] * (int32_t)q[56]; tmp1 = (int32_t)in[40] * (int32_t)q[40]; tmp2 = (int32_t)in[24] * (int32_t)q[24]; tmp3 = (int32_t)in[8] * (int32_t)q[8];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        int64_t z4 = tmp1 + tmp3;
        const int64_t z5 = (z3 + z4) * F_1_175875602;
        tmp0 *= F_0_298631336; tmp1 *= F_2_053119869; tmp2 *= F_3_072711026; tmp3 *= F_1_501321110;
        z1 *= -F_0_899976223; z2 *= -F_2_562915447; z3 *= -F_1_961570560; z4 *= -F_0_390180644;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        w[0] = descale(tmp10 + tmp3, CONST_BITS - PASS1_BITS);  w[56] = descale(tmp10 - tmp3, CONST_BITS - PASS1_BITS);
        w[8] = descale(tmp11 + tmp2, CONST_BITS - PASS1_BITS);  w[48] = descale(tmp11 - tmp2, CONST_BITS - PASS1_BITS);
        w[16] = descale(tmp12 + tmp1, CONST_BITS - PASS1_BITS); w[40] = descale(tmp12 - tmp1, CONST_BITS - PASS1_BITS);
        w[24] = descale(tmp13 + tmp0, CONST_BITS - PASS1_BITS); w[32] = descale(tmp13 - tmp0, CONST_BITS - PASS1_BITS);
    }
    for (int r = 0; r < 8; ++r) {
        const int32_t* w = ws + 8 * r; uint8_t* o = out + (size_t)r * pitch;
        if (w[1] == 0 && w[2] == 0 && w[3] == 0 && w[4] == 0 && w[5] == 0 && w[6] == 0 && w[7] == 0) {
            const uint8_t dc = range_limit(descale(w[0], PASS1_BITS + 3));
            for (int c = 0; c < 8; ++c) o[c] = dc;
            continue;
        }
        int64_t z2 = w[2], z3 = w[6];
        int64_t z1 = (z2 + z3) * F_0_541196100;
        int64_t tmp2 = z1 + z3 * (-F_1_847759065), tmp3 = z1 + z2 * F_0_765366865;
        int64_t tmp0 = ((int64_t)w[0] + w[4]) * (1 << CONST_BITS), tmp1 = ((int64_t)w[0] - w[4]) * (1 << CONST_BITS);
        const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        int64_t z4 = tmp1 + tmp3;
        const int64_t z5 = (z3 + z4) * F_1_175875602;
        tmp0 *= F_0_298631336; tmp1 *= F_2_053119869; tmp2 *= F_3_072711026; tmp3 *= F_1_501321110;
        z1 *= -F_0_899976223; z2 *= -F_2_562915447; z3 *= -F_1_961570560; z4 *= -F_0_390180644;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        constexpr int S = CONST_BITS + PASS1_BITS + 3;
        o[0] = range_limit(descale(tmp10 + tmp3, S)); o[7] = range_limit(descale(tmp10 - tmp3, S));
        o[1] = range_limit(descale(tmp11 + tmp2, S)); o[6] = range_limit(descale(tmp11 - tmp2, S));
        o[2] = range_limit(descale(tmp12 + tmp1, S)); o[5] = range_limit(descale(tmp12 - tmp1, S));
        o[3] = range_limit(descale(tmp13 + tmp0, S)); o[4] = range_limit(descale(tmp13 - tmp0, S));
    }
}

struct Component { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0; int pred = 0; };

bool fail(std::string* why, const char* msg) { if (why) *why = msg; return false; }

}  // namespace

bool decode_jpeg_gray(const uint8_t* d, size_t size, GrayImage& out, std::string* why)
{
    if (!looks_like_jpeg(d, size)) return fail(why, "not a JPEG stream (no SOI)");
    uint16_t quant[4][64]; bool have_q[4] = {false, false, false, false};
    HuffTable dc[4], ac[4];
    Component comp[4];
    int ncomp = 0, X = 0, Y = 0, hmax = 1, vmax = 1, restart_interval = 0;
    bool have_frame = false, decoded_luma = false;
    std::vector<uint8_t> plane;            // component 0, padded to whole blocks / MCUs
    int plane_w = 0, plane_h = 0;
    size_t pos = 2;
    auto u16 = [&](size_t o) { return (int)((d[o] << 8) | d[o + 1]); };
    for (;;) {
        // next marker (skip anything that is not FF, then fill FFs)
        while (pos < size && d[pos] != 0xFF) ++pos;
        while (pos < size && d[pos] == 0xFF) ++pos;
        if (pos >= size) break;
        const int m = d[pos++];
        if (m == 0xD9) break;                                                          // EOI
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7) || m == 0x00) continue;               // stand-alone
        if (pos + 2 > size) return fail(why, "truncated marker segment");
        const int len = u16(pos);
        if (len < 2 || pos + (size_t)len > size) return fail(why, "bad marker segment length");
        const uint8_t* s = d + pos + 2; const int n = len - 2;
        if (m == 0xDB) {                                                               // DQT
            int o = 0;
            while (o < n) {
                const int pq = s[o] >> 4, tq = s[o] & 15; ++o;
                if (tq > 3 || pq > 1 || o + 64 * (pq + 1) > n) return fail(why, "bad quantisation table");
                for (int k = 0; k < 64; ++k) { quant[tq][kZigzag[k]] = pq ? (uint16_t)((s[o] << 8) | s[o + 1]) : s[o]; o += pq + 1; }
                have_q[tq] = true;
            }
        } else if (m == 0xC4) {                                                        // DHT
            int o = 0;
            while (o < n) {
                if (o + 17 > n) return fail(why, "bad Huffman table");
                const int tc = s[o] >> 4, th = s[o] & 15; ++o;
                if (tc > 1 || th > 3) return fail(why, "bad Huffman table id");
                HuffTable& t = tc ? ac[th] : dc[th];
                int total = 0;
                for (int l = 1; l <= 16; ++l) { t.bits[l] = s[o + l - 1]; total += t.bits[l]; }
                o += 16;
                if (total > 256 || o + total > n) return fail(why, "bad Huffman table size");
                std::memcpy(t.vals, s + o, (size_t)total); o += total;
                if (!t.build()) return fail(why, "inconsistent Huffman table");
                t.present = true;
            }
        } else if (m == 0xC0 || m == 0xC1) {                                           // SOF0 / SOF1: Huffman, sequential
            if (have_frame) return fail(why, "second frame header");
            if (n < 6) return fail(why, "bad frame header");
            if (s[0] != 8) return fail(why, "only 8-bit samples are supported");
            Y = u16(pos + 3); X = u16(pos + 5); ncomp = s[5];
            if (X <= 0 || Y <= 0 || (ncomp != 1 && ncomp != 3) || n < 6 + 3 * ncomp) return fail(why, "unsupported frame (size / component count)");
            if ((size_t)X * (size_t)Y > (size_t)1 << 28) return fail(why, "frame too large");
            for (int i = 0; i < ncomp; ++i) {
                comp[i].id = s[6 + 3 * i]; comp[i].h = s[7 + 3 * i] >> 4; comp[i].v = s[7 + 3 * i] & 15; comp[i].tq = s[8 + 3 * i];
                if (comp[i].h < 1 || comp[i].h > 4 || comp[i].v < 1 || comp[i].v > 4 || comp[i].tq > 3) return fail(why, "bad component description");
                hmax = std::max(hmax, comp[i].h); vmax = std::max(vmax, comp[i].v);
            }
            if (ncomp == 1) { comp[0].h = comp[0].v = 1; hmax = vmax = 1; }             // a single component is never interleaved
            // the grey output is component 0's plane as it is coded: a file whose first component is SUBSAMPLED against another one
            // (luma 1x1 beside chroma 2x2, which no camera or encoder of this code base writes) would need libjpeg's upsampling
            if (comp[0].h != hmax || comp[0].v != vmax) return fail(why, "first component is subsampled (not supported)");
            const int mcux = (X + 8 * hmax - 1) / (8 * hmax), mcuy = (Y + 8 * vmax - 1) / (8 * vmax);
            plane_w = mcux * comp[0].h * 8; plane_h = mcuy * comp[0].v * 8;
            if (plane_w < X || plane_h < Y) return fail(why, "frame geometry is inconsistent");
            plane.assign((size_t)plane_w * plane_h, 0);
            have_frame = true;
        } else if (m == 0xC2 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC) || m == 0xC3) {
            return fail(why, m == 0xC2 ? "progressive JPEG is not supported (baseline only)" : "unsupported JPEG process (arithmetic / lossless / hierarchical)");
        } else if (m == 0xDD) {                                                        // DRI
            if (n < 2) return fail(why, "bad restart interval");
            restart_interval = u16(pos + 2);
        } else if (m == 0xDA) {                                                        // SOS + entropy-coded data
            if (!have_frame) return fail(why, "scan before the frame header");
            if (n < 1) return fail(why, "bad scan header");
            const int ns = s[0];
            if (ns < 1 || ns > ncomp || n < 1 + 2 * ns + 3) return fail(why, "bad scan header");
            int idx[4];
            for (int i = 0; i < ns; ++i) {
                int ci = -1;
                for (int c = 0; c < ncomp; ++c) if (comp[c].id == s[1 + 2 * i]) ci = c;
                if (ci < 0) return fail(why, "scan names an unknown component");
                for (int k = 0; k < i; ++k) if (idx[k] == ci) return fail(why, "scan names a component twice");
                if (ci == 0 && decoded_luma) return fail(why, "second scan of the first component in a sequential file");
                comp[ci].td = s[2 + 2 * i] >> 4; comp[ci].ta = s[2 + 2 * i] & 15;
                if (comp[ci].td > 3 || comp[ci].ta > 3 || !dc[comp[ci].td].present || !ac[comp[ci].ta].present || !have_q[comp[ci].tq]) return fail(why, "scan uses a missing table");
                idx[i] = ci;
            }
            BitReader br{d + pos + (size_t)len, d + size};
            for (int c = 0; c < ncomp; ++c) comp[c].pred = 0;
            int16_t block[64];
            auto decode_block = [&](Component& c, bool keep, int bx, int by) -> bool {
                std::memset(block, 0, sizeof(block));
                int sym;
                if (!decode_symbol(br, dc[c.td], sym) || sym > 11) return false;
                c.pred += sym ? extend(br.get(sym), sym) : 0;
                block[0] = (int16_t)c.pred;
                for (int k = 1; k < 64;) {
                    if (!decode_symbol(br, ac[c.ta], sym)) return false;
                    const int r = sym >> 4, sz = sym & 15;
                    if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
                    k += r;
                    if (k > 63) return false;
                    block[kZigzag[k]] = (int16_t)extend(br.get(sz), sz);
                    ++k;
                }
                if (keep) idct_islow(block, quant[c.tq], plane.data() + (size_t)by * 8 * plane_w + (size_t)bx * 8, plane_w);
                return true;
            };
            auto restart = [&]() -> bool {
                // the bit reader stops at the marker: it must be an RSTn; then byte alignment and fresh predictions
                br.fill();
                if (br.marker < 0xD0 || br.marker > 0xD7) return false;
                br.restart();
                for (int c = 0; c < ncomp; ++c) comp[c].pred = 0;
                return true;
            };
            if (ns == 1) {                                                             // non-interleaved: the component's own block raster
                Component& c = comp[idx[0]];
                const int cw = (X * c.h + hmax - 1) / hmax, chh = (Y * c.v + vmax - 1) / vmax;
                const int bw = (cw + 7) / 8, bh = (chh + 7) / 8;
                int count = 0;
                for (int by = 0; by < bh; ++by)
                    for (int bx = 0; bx < bw; ++bx) {
                        if (restart_interval && count && count % restart_interval == 0 && !restart()) return fail(why, "missing restart marker");
                        const bool keep = idx[0] == 0 && (bx + 1) * 8 <= plane_w && (by + 1) * 8 <= plane_h;
                        if (!decode_block(c, keep, bx, by)) return fail(why, "corrupt entropy-coded data");
                        ++count;
                    }
            } else {
                const int mcux = (X + 8 * hmax - 1) / (8 * hmax), mcuy = (Y + 8 * vmax - 1) / (8 * vmax);
                int count = 0;
                for (int my = 0; my < mcuy; ++my)
                    for (int mx = 0; mx < mcux; ++mx) {
                        if (restart_interval && count && count % restart_interval == 0 && !restart()) return fail(why, "missing restart marker");
                        for (int i = 0; i < ns; ++i) {
                            Component& c = comp[idx[i]];
                            for (int v = 0; v < c.v; ++v)
                                for (int h = 0; h < c.h; ++h)
                                    if (!decode_block(c, idx[i] == 0, mx * c.h + h, my * c.v + v)) return fail(why, "corrupt entropy-coded data");
                        }
                        ++count;
                    }
            }
            for (int i = 0; i < ns; ++i) if (idx[i] == 0) decoded_luma = true;
            // continue behind the entropy-coded data: at the marker the reader stopped at, or scan for the next one
            size_t q = (size_t)(br.p - d);
            pos = q >= 2 ? q - 2 : q;
            if (br.marker) { pos = q - 2; continue; }
            continue;
        }
        pos += (size_t)len;
    }
    if (!have_frame || !decoded_luma) return fail(why, "no image data");
    if (plane_w < X || plane_h < Y || plane.size() < (size_t)plane_w * (size_t)Y) return fail(why, "frame geometry is inconsistent");
    out.width = X; out.height = Y;
    out.pixels.resize((size_t)X * Y);
    for (int y = 0; y < Y; ++y) std::memcpy(&out.pixels[(size_t)y * X], &plane[(size_t)y * plane_w], (size_t)X);
    return true;
}

}  // namespace LpSlam

namespace LpSlam {
namespace {

// ---- encoder: what cv::imencode(".jpg", grey) asks libjpeg for -- one component, baseline, the standard (Annex K) tables, quality 95
const uint8_t kStdLumQuant[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                                  18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const uint8_t kDcLumBits[17] = {0, 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kDcLumVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kAcLumBits[17] = {0, 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t kAcLumVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1,
    0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a,
    0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3,
    0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

struct EncTable { uint16_t code[256]; uint8_t len[256]; };
void make_enc_table(const uint8_t* bits, const uint8_t* vals, EncTable& t)
{
    std::memset(&t, 0, sizeof(t));
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
        for (int i = 0; i < bits[l]; ++i, ++k, ++code) { t.code[vals[k]] = (uint16_t)code; t.len[vals[k]] = (uint8_t)l; }
        code <<= 1;
    }
}

// libjpeg's jpeg_fdct_islow on an 8 x 8 block of level-shifted samples (the result is 8 x the DCT)
void fdct_islow(int32_t* data)
{
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = 0; i < 8; ++i) {
            int32_t* d = pass == 0 ? data + 8 * i : data + i;
            const int st = pass == 0 ? 1 : 8;
            const int64_t tmp0 = d[0] + d[7 * st], tmp7 = d[0] - d[7 * st], tmp1 = d[st] + d[6 * st], tmp6 = d[st] - d[6 * st];
            const int64_t tmp2 = d[2 * st] + d[5 * st], tmp5 = d[2 * st] - d[5 * st], tmp3 = d[3 * st] + d[4 * st], tmp4 = d[3 * st] - d[4 * st];
            const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            const int sh = pass == 0 ? CONST_BITS - PASS1_BITS : CONST_BITS + PASS1_BITS;
            if (pass == 0) { d[0] = (int32_t)((tmp10 + tmp11) * (1 << PASS1_BITS)); d[4 * st] = (int32_t)((tmp10 - tmp11) * (1 << PASS1_BITS)); }
            else { d[0] = descale(tmp10 + tmp11, PASS1_BITS); d[4 * st] = descale(tmp10 - tmp11, PASS1_BITS); }
            int64_t z1 = (tmp12 + tmp13) * F_0_541196100;
            d[2 * st] = descale(z1 + tmp13 * F_0_765366865, sh);
            d[6 * st] = descale(z1 + tmp12 * (-F_1_847759065), sh);
            z1 = tmp4 + tmp7;
            int64_t z2 = tmp5 + tmp6, z3 = tmp4 + tmp6, z4 = tmp5 + tmp7;
            const int64_t z5 = (z3 + z4) * F_1_175875602;
            const int64_t t4 = tmp4 * F_0_298631336, t5 = tmp5 * F_2_053119869, t6 = tmp6 * F_3_072711026, t7 = tmp7 * F_1_501321110;
            z1 *= -F_0_899976223; z2 *= -F_2_562915447; z3 *= -F_1_961570560; z4 *= -F_0_390180644;
            z3 += z5; z4 += z5;
            d[7 * st] = descale(t4 + z1 + z3, sh); d[5 * st] = descale(t5 + z2 + z4, sh);
            d[3 * st] = descale(t6 + z2 + z3, sh); d[st] = descale(t7 + z1 + z4, sh);
        }
    }
}

struct BitWriter {
    std::vector<uint8_t>& out; uint32_t acc = 0; int n = 0;
    void put(uint32_t code, int len)
    {
        acc = (acc << len) | (code & ((1u << len) - 1)); n += len;
        while (n >= 8) { const uint8_t b = (uint8_t)(acc >> (n - 8)); out.push_back(b); if (b == 0xFF) out.push_back(0); n -= 8; }
    }
    void flush() { if (n) put(0x7F, 8 - n); }             // pad with ones
};

}  // namespace

bool encode_jpeg_gray(const GrayImage& img, int quality, std::vector<uint8_t>& out)
{
    if (img.width <= 0 || img.height <= 0 || img.width > 65535 || img.height > 65535 || img.pixels.size() < (size_t)img.width * img.height) return false;
    quality = std::min(100, std::max(1, quality));
    const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;          // jpeg_quality_scaling
    uint8_t q[64];
    for (int i = 0; i < 64; ++i) { long t = ((long)kStdLumQuant[i] * scale + 50) / 100; q[i] = (uint8_t)std::min(255L, std::max(1L, t)); }      // force_baseline
    EncTable dct, act;
    make_enc_table(kDcLumBits, kDcLumVals, dct); make_enc_table(kAcLumBits, kAcLumVals, act);
    out.clear();
    auto put16 = [&](int v) { out.push_back((uint8_t)(v >> 8)); out.push_back((uint8_t)v); };
    auto marker = [&](int m) { out.push_back(0xFF); out.push_back((uint8_t)m); };
    marker(0xD8);
    marker(0xE0); put16(16); for (char c : {'J', 'F', 'I', 'F', '\0'}) out.push_back((uint8_t)c);
    out.push_back(1); out.push_back(1); out.push_back(0); put16(1); put16(1); out.push_back(0); out.push_back(0);      // JFIF 1.01, no density unit, 1:1
    marker(0xDB); put16(67); out.push_back(0); for (int k = 0; k < 64; ++k) out.push_back(q[kZigzag[k]]);
    marker(0xC0); put16(11); out.push_back(8); put16(img.height); put16(img.width); out.push_back(1); out.push_back(1); out.push_back(0x11); out.push_back(0);
    marker(0xC4); put16(2 + 1 + 16 + 12); out.push_back(0x00); for (int l = 1; l <= 16; ++l) out.push_back(kDcLumBits[l]); for (uint8_t v : kDcLumVals) out.push_back(v);
    marker(0xC4); put16(2 + 1 + 16 + 162); out.push_back(0x10); for (int l = 1; l <= 16; ++l) out.push_back(kAcLumBits[l]); for (uint8_t v : kAcLumVals) out.push_back(v);
    marker(0xDA); put16(8); out.push_back(1); out.push_back(1); out.push_back(0x00); out.push_back(0); out.push_back(63); out.push_back(0);
    BitWriter bw{out};
    const int bw_n = (img.width + 7) / 8, bh_n = (img.height + 7) / 8;
    int pred = 0;
    for (int by = 0; by < bh_n; ++by)
        for (int bx = 0; bx < bw_n; ++bx) {
            int32_t blk[64];
            for (int r = 0; r < 8; ++r) {
                const int y = std::min(8 * by + r, img.height - 1);                // edge blocks: the last row / column repeated (libjpeg's edge expansion)
                for (int c = 0; c < 8; ++c) blk[8 * r + c] = (int32_t)img.pixels[(size_t)y * img.width + std::min(8 * bx + c, img.width - 1)] - 128;
            }
            fdct_islow(blk);
            int16_t zz[64];
            for (int k = 0; k < 64; ++k) {
                const int i = kZigzag[k];
                const int32_t qv = (int32_t)q[i] << 3;                            // the transform's output is scaled by 8
                int32_t t = blk[i];
                if (t < 0) { t = -t; t += qv >> 1; t = t >= qv ? t / qv : 0; t = -t; }
                else { t += qv >> 1; t = t >= qv ? t / qv : 0; }
                zz[k] = (int16_t)t;
            }
            auto category = [](int v) { int a = v < 0 ? -v : v, s = 0; while (a) { ++s; a >>= 1; } return s; };
            const int diff = zz[0] - pred; pred = zz[0];
            int s2 = category(diff);
            bw.put(dct.code[s2], dct.len[s2]);
            if (s2) bw.put((uint32_t)(diff < 0 ? diff - 1 : diff), s2);
            int run = 0;
            for (int k = 1; k < 64; ++k) {
                const int v = zz[k];
                if (v == 0) { ++run; continue; }
                while (run > 15) { bw.put(act.code[0xF0], act.len[0xF0]); run -= 16; }
                s2 = category(v);
                bw.put(act.code[(run << 4) | s2], act.len[(run << 4) | s2]);
                bw.put((uint32_t)(v < 0 ? v - 1 : v), s2);
                run = 0;
            }
            if (run) bw.put(act.code[0], act.len[0]);
        }
    bw.flush();
    marker(0xD9);
    return true;
}

}  // namespace LpSlam

// tests: encode a grey image; returns the number of bytes written, 0 when the buffer is too small or the image is not encodable
extern "C" __attribute__((visibility("default"))) size_t lpslam_jpeg_encode_gray(const uint8_t* pixels, int w, int h, int quality, uint8_t* out, size_t cap)
{
    LpSlam::GrayImage img; img.width = w; img.height = h; img.pixels.assign(pixels, pixels + (size_t)w * h);
    std::vector<uint8_t> buf;
    if (!LpSlam::encode_jpeg_gray(img, quality, buf) || buf.size() > cap) return 0;
    std::memcpy(out, buf.data(), buf.size());
    return buf.size();
}

// tests: decode into a caller buffer; returns 0 on success, 1 when the buffer is too small (w / h are set), 2 on a decoding error
extern "C" __attribute__((visibility("default"))) int lpslam_jpeg_decode_gray(const uint8_t* data, size_t size, uint8_t* out, size_t cap, int* w, int* h)
{
    LpSlam::GrayImage img;
    if (!LpSlam::decode_jpeg_gray(data, size, img, nullptr)) return 2;
    if (w) *w = img.width;
    if (h) *h = img.height;
    if (img.pixels.size() > cap) return 1;
    std::memcpy(out, img.pixels.data(), img.pixels.size());
    return 0;
}
