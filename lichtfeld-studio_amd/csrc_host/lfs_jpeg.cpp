// JPEG decoder of liblfs_io.so: 8-bit Huffman JPEG (baseline, extended sequential and progressive DCT), 1 or 3 components, any sampling
// factors with h, v in {1, 2}, restart intervals -> 8-bit RGB. Part of SURVEY.md §8f row 4: the reference reads its training images through
// OpenImageIO (src/core/image_io.cpp:112-270), which hands JPEG to libjpeg(-turbo) with the library defaults; "identical COLMAP inputs"
// therefore means the libjpeg default pipeline, restated here from the IJG / libjpeg-turbo sources it is published in (third-party, not under
// /root/reference):
//     inverse DCT            jidctint.c   "slow but accurate" integer IDCT (JDCT_ISLOW, the default), CONST_BITS 13, PASS1_BITS 2
//     chroma upsampling      jdsample.c   fancy (triangle) upsampling h2v1 / h2v2 (do_fancy_upsampling = TRUE, the default); the replicated
//                                         context rows of jdmainct.c at the top and bottom image edge
//     colour conversion      jdcolor.c    YCbCr -> RGB with the 16-bit fixed-point tables
// so the output is bit-identical to libjpeg-turbo (what Pillow links; tests/test_loader_io.py compares byte for byte).
// Not supported (lfs_image_load_rgb8 reports LFS_IO_E_UNSUPPORTED and the host layer may fall back): arithmetic coding, 12-bit, lossless,
// CMYK / YCCK, sampling factors above 2.
#include "../../include/lfs_io.h"

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace lfs_jpeg {

struct Unsupported : std::runtime_error { using std::runtime_error::runtime_error; };
struct Corrupt : std::runtime_error { using std::runtime_error::runtime_error; };

static const uint8_t kZigzag[64 + 16] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                                         6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                                         39, 46, 53, 60, 61, 54, 47, 55, 62, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct Huff {
    bool defined = false;
    uint8_t bits[17] = {0}, vals[256] = {0};
    // canonical decode tables (Annex F.2.2.3)
    int32_t mincode[17], maxcode[18], valptr[17];
    uint8_t look_n[512]; uint8_t look_v[512]; // 9-bit fast path
    void build() {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k; mincode[l] = code;
            k += bits[l]; code += bits[l];
            if (code > (1 << l)) throw Corrupt("JPEG: over-subscribed Huffman table"); // more codes of length l than the prefix tree has room for
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        std::memset(look_n, 0, sizeof look_n);
        code = 0; k = 0;
        for (int l = 1; l <= 9; ++l) {
            for (int i = 0; i < bits[l]; ++i, ++k, ++code) {
                const int lo = code << (9 - l), n = 1 << (9 - l);
                for (int j = 0; j < n; ++j) { look_n[lo + j] = (uint8_t)l; look_v[lo + j] = vals[k]; }
            }
            code <<= 1;
        }
        defined = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int blocks_w = 0, blocks_h = 0;        // allocated blocks (whole MCUs)
    int width = 0, height = 0;             // downsampled_width / downsampled_height (samples that carry image content)
    std::vector<int16_t> coef;             // [blocks_h][blocks_w][64], natural order
    std::vector<uint8_t> plane;            // [blocks_h * 8][blocks_w * 8]
    int dc_pred = 0;
};

class BitReader {
  public:
    BitReader(const uint8_t* p, const uint8_t* end) : p_(p), end_(end) {}
    const uint8_t* pos() const { return p_; }
    void reset(const uint8_t* p) { p_ = p; acc_ = 0; n_ = 0; marker_ = 0; }
    int marker() const { return marker_; }
    inline void fill() {
        while (n_ <= 24) {
            int b = 0;
            if (!marker_ && p_ < end_) {
                b = *p_++;
                if (b == 0xFF) {
                    int c = p_ < end_ ? *p_ : 0xD9;
                    while (c == 0xFF && p_ + 1 < end_) { ++p_; c = *p_; } // fill bytes
                    if (c == 0) ++p_;                                      // stuffed zero
                    else { marker_ = c; ++p_; b = 0; }                     // a marker ends the entropy-coded segment: feed zeros
                }
            }
            acc_ |= (uint32_t)b << (24 - n_);
            n_ += 8;
        }
    }
    inline int peek(int n) { if (n_ < n) fill(); return (int)(acc_ >> (32 - n)); }
    inline void skip(int n) { acc_ <<= n; n_ -= n; }
    inline int bits(int n) { if (!n) return 0; if (n_ < n) fill(); const int v = (int)(acc_ >> (32 - n)); skip(n); return v; }
    inline int bit() { return bits(1); }
    inline int receive_extend(int s) { // F.2.2.1
        if (!s) return 0;
        const int v = bits(s);
        return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
    }
    inline int decode(const Huff& h) {
        if (n_ < 16) fill();
        const int look = (int)(acc_ >> 23);
        if (h.look_n[look]) { skip(h.look_n[look]); return h.look_v[look]; }
        int code = (int)(acc_ >> 22), l = 10; // 10 bits
        while (l <= 16 && code > h.maxcode[l]) { ++l; code = (int)(acc_ >> (32 - l)); }
        if (l > 16) throw Corrupt("JPEG: bad Huffman code");
        skip(l);
        return h.vals[(h.valptr[l] + code - h.mincode[l]) & 255];
    }

  private:
    const uint8_t* p_; const uint8_t* end_;
    uint32_t acc_ = 0; int n_ = 0; int marker_ = 0;
};

struct Decoder {
    const uint8_t* data; size_t size;
    int width = 0, height = 0, ncomp = 0, hmax = 1, vmax = 1, mcus_x = 0, mcus_y = 0;
    bool progressive = false, jfif = false, adobe = false; int adobe_transform = -1;
    uint16_t qt[4][64]; bool qt_defined[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    Component comp[3];
    int restart_interval = 0;
    int eobrun = 0;

    Decoder(const uint8_t* d, size_t n) : data(d), size(n) {}

    static int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

    void parse_dqt(const uint8_t* p, int len) {
        while (len > 0) {
            const int pq = p[0] >> 4, tq = p[0] & 15;
            if (tq > 3 || pq > 1) throw Corrupt("JPEG: bad DQT");
            const int need = 1 + 64 * (pq + 1);
            if (len < need) throw Corrupt("JPEG: short DQT");
            for (int i = 0; i < 64; ++i) qt[tq][kZigzag[i]] = pq ? (uint16_t)be16(p + 1 + 2 * i) : p[1 + i];
            qt_defined[tq] = true;
            p += need; len -= need;
        }
    }
    void parse_dht(const uint8_t* p, int len) {
        while (len > 0) {
            if (len < 17) throw Corrupt("JPEG: short DHT");
            const int tc = p[0] >> 4, th = p[0] & 15;
            if (tc > 1 || th > 3) throw Corrupt("JPEG: bad DHT");
            Huff& h = tc ? ac[th] : dc[th];
            int total = 0;
            h.bits[0] = 0;
            for (int i = 1; i <= 16; ++i) { h.bits[i] = p[i]; total += p[i]; }
            if (total > 256 || len < 17 + total) throw Corrupt("JPEG: bad DHT counts");
            std::memset(h.vals, 0, sizeof h.vals);
            std::memcpy(h.vals, p + 17, (size_t)total);
            h.build();
            p += 17 + total; len -= 17 + total;
        }
    }
    void parse_sof(const uint8_t* p, int len, int marker) {
        if (marker == 0xC3 || marker == 0xC5 || marker == 0xC6 || marker == 0xC7 || marker >= 0xC9) throw Unsupported("JPEG: lossless / hierarchical / arithmetic coding is not supported");
        progressive = marker == 0xC2;
        if (len < 6 || p[0] != 8) throw Unsupported("JPEG: only 8-bit precision is supported");
        height = be16(p + 1); width = be16(p + 3); ncomp = p[5];
        if (width <= 0 || height <= 0) throw Corrupt("JPEG: empty image");
        if ((uint64_t)width * (uint64_t)height > (uint64_t(1) << 28)) throw Corrupt("JPEG: implausible dimensions"); // 268 Mpixel: the coefficient planes alone would take > 1.5 GB
        if (ncomp != 1 && ncomp != 3) throw Unsupported("JPEG: only 1- and 3-component images are supported");
        if (len < 6 + 3 * ncomp) throw Corrupt("JPEG: short SOF");
        for (int i = 0; i < ncomp; ++i) {
            Component& c = comp[i];
            c.id = p[6 + 3 * i]; c.h = p[7 + 3 * i] >> 4; c.v = p[7 + 3 * i] & 15; c.tq = p[8 + 3 * i];
            if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2 || c.tq > 3) throw Unsupported("JPEG: sampling factors above 2 are not supported");
            hmax = c.h > hmax ? c.h : hmax; vmax = c.v > vmax ? c.v : vmax;
        }
        if (ncomp == 1) { comp[0].h = comp[0].v = 1; hmax = vmax = 1; } // a single component is never subsampled (B.2.3)
        mcus_x = (width + 8 * hmax - 1) / (8 * hmax); mcus_y = (height + 8 * vmax - 1) / (8 * vmax);
        for (int i = 0; i < ncomp; ++i) {
            Component& c = comp[i];
            c.blocks_w = mcus_x * c.h; c.blocks_h = mcus_y * c.v;
            c.width = (width * c.h + hmax - 1) / hmax; c.height = (height * c.v + vmax - 1) / vmax;
            c.coef.assign((size_t)c.blocks_w * c.blocks_h * 64, 0);
        }
    }

    // ---- entropy decoding -------------------------------------------------------------------------------------------
    void block_baseline(BitReader& br, Component& c, int16_t* blk) {
        const Huff& hd = dc[c.td]; const Huff& ha = ac[c.ta];
        const int s = br.decode(hd);
        if (s > 16) throw Corrupt("JPEG: bad DC difference category");
        c.dc_pred = (int)((unsigned)c.dc_pred + (unsigned)br.receive_extend(s)); // wraps on corrupt streams instead of overflowing
        blk[0] = (int16_t)c.dc_pred;
        for (int k = 1; k < 64;) {
            const int rs = br.decode(ha), r = rs >> 4, ss = rs & 15;
            if (ss) { k += r; blk[kZigzag[k]] = (int16_t)br.receive_extend(ss); ++k; }
            else { if (r != 15) break; k += 16; }
        }
    }
    void block_dc_first(BitReader& br, Component& c, int16_t* blk, int al) {
        const int s = br.decode(dc[c.td]);
        if (s > 16) throw Corrupt("JPEG: bad DC difference category");
        c.dc_pred = (int)((unsigned)c.dc_pred + (unsigned)br.receive_extend(s));
        blk[0] = (int16_t)((unsigned)c.dc_pred << al);
    }
    void block_dc_refine(BitReader& br, int16_t* blk, int al) { if (br.bit()) blk[0] |= (int16_t)(1 << al); }
    void block_ac_first(BitReader& br, Component& c, int16_t* blk, int ss, int se, int al) {
        if (eobrun) { --eobrun; return; }
        const Huff& ha = ac[c.ta];
        for (int k = ss; k <= se;) {
            const int rs = br.decode(ha), r = rs >> 4, s = rs & 15;
            if (s) { k += r; blk[kZigzag[k]] = (int16_t)(br.receive_extend(s) * (1 << al)); ++k; }
            else if (r == 15) k += 16;
            else { eobrun = (1 << r) - 1; if (r) eobrun += br.bits(r); break; }
        }
    }
    void block_ac_refine(BitReader& br, Component& c, int16_t* blk, int ss, int se, int al) { // G.1.2.3
        const int p1 = 1 << al, m1 = -1 * (1 << al);
        const Huff& ha = ac[c.ta];
        int k = ss;
        if (!eobrun) {
            for (; k <= se;) {
                const int rs = br.decode(ha);
                int r = rs >> 4, s = rs & 15, val = 0;
                if (s) { val = br.bit() ? p1 : m1; }
                else if (r != 15) { eobrun = 1 << r; if (r) eobrun += br.bits(r); break; }
                // skip r zero-history coefficients, refining the non-zero ones passed on the way
                for (; k <= se; ++k) {
                    int16_t& co = blk[kZigzag[k]];
                    if (co) { if (br.bit() && !(co & p1)) co = (int16_t)(co >= 0 ? co + p1 : co + m1); }
                    else { if (--r < 0) break; }
                }
                if (s && k <= se) blk[kZigzag[k]] = (int16_t)val;
                ++k;
            }
        }
        if (eobrun) { // refine the rest of the band
            for (; k <= se; ++k) {
                int16_t& co = blk[kZigzag[k]];
                if (co && br.bit() && !(co & p1)) co = (int16_t)(co >= 0 ? co + p1 : co + m1);
            }
            --eobrun;
        }
    }

    const uint8_t* decode_scan(const uint8_t* p, int len) {
        const int ns = p[0];
        if (ns < 1 || ns > ncomp || len < 4 + 2 * ns) throw Corrupt("JPEG: bad SOS");
        Component* sc[3];
        for (int i = 0; i < ns; ++i) {
            sc[i] = nullptr;
            for (int j = 0; j < ncomp; ++j) if (comp[j].id == p[1 + 2 * i]) sc[i] = &comp[j];
            if (!sc[i]) throw Corrupt("JPEG: SOS names an unknown component");
            sc[i]->td = p[2 + 2 * i] >> 4; sc[i]->ta = p[2 + 2 * i] & 15;
            if (sc[i]->td > 3 || sc[i]->ta > 3) throw Corrupt("JPEG: bad table selector");
        }
        const int ss = p[1 + 2 * ns], se = p[2 + 2 * ns], ah = p[3 + 2 * ns] >> 4, al = p[3 + 2 * ns] & 15;
        if (progressive ? (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1) || al > 13) : false) throw Corrupt("JPEG: bad progressive scan parameters");
        for (int i = 0; i < ns; ++i) {
            if ((!progressive || ss == 0) && !(progressive && ah) && !dc[sc[i]->td].defined) throw Corrupt("JPEG: missing DC Huffman table");
            if ((!progressive || ss > 0) && !ac[sc[i]->ta].defined) throw Corrupt("JPEG: missing AC Huffman table");
        }
        BitReader br(p + len, data + size);
        for (int i = 0; i < ncomp; ++i) comp[i].dc_pred = 0;
        eobrun = 0;
        auto do_block = [&](Component& c, int bx, int by) {
            int16_t* blk = c.coef.data() + ((size_t)by * c.blocks_w + bx) * 64;
            if (!progressive) block_baseline(br, c, blk);
            else if (ss == 0) { if (ah == 0) block_dc_first(br, c, blk, al); else block_dc_refine(br, blk, al); }
            else { if (ah == 0) block_ac_first(br, c, blk, ss, se, al); else block_ac_refine(br, c, blk, ss, se, al); }
        };
        int restarts_left = restart_interval, next_rst = 0;
        auto restart_check = [&]() {
            if (!restart_interval) return;
            if (--restarts_left > 0) return;
            // expect RSTn: skip to it
            br.fill();
            const uint8_t* q = br.pos();
            if (br.marker() >= 0xD0 && br.marker() <= 0xD7) { /* consumed by the reader */ }
            else if (br.marker()) return; // some other marker: the scan has ended (no RST follows the last interval)
            else {
                while (q + 1 < data + size && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) {
                    if (q[0] == 0xFF && q[1] != 0 && q[1] != 0xFF) return; // some other marker: the scan ends
                    ++q;
                }
                if (q + 1 >= data + size) return;
                q += 2;
            }
            br.reset(q);
            (void)next_rst;
            for (int i = 0; i < ncomp; ++i) comp[i].dc_pred = 0;
            eobrun = 0;
            restarts_left = restart_interval;
        };
        if (ns == 1) { // non-interleaved: the component's own block raster, ceil(size / 8) blocks
            Component& c = *sc[0];
            const int bw = (c.width + 7) / 8, bh = (c.height + 7) / 8;
            for (int by = 0; by < bh; ++by)
                for (int bx = 0; bx < bw; ++bx) { do_block(c, bx, by); restart_check(); }
        } else {
            for (int my = 0; my < mcus_y; ++my)
                for (int mx = 0; mx < mcus_x; ++mx) {
                    for (int i = 0; i < ns; ++i)
                        for (int v = 0; v < sc[i]->v; ++v)
                            for (int h = 0; h < sc[i]->h; ++h) do_block(*sc[i], mx * sc[i]->h + h, my * sc[i]->v + v);
                    restart_check();
                }
        }
        // position after the entropy-coded segment: the reader stopped at a marker (or at the end)
        br.fill();
        const uint8_t* q = br.pos();
        if (br.marker()) return q - 2;
        while (q + 1 < data + size && !(q[0] == 0xFF && q[1] != 0 && q[1] != 0xFF && !(q[1] >= 0xD0 && q[1] <= 0xD7))) ++q;
        return q;
    }

    // ---- jidctint.c --------------------------------------------------------------------------------------------------
    static inline int descale(int64_t x, int n) { return (int)((x + ((int64_t)1 << (n - 1))) >> n); }
    static inline uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
    static void idct_islow(const int16_t* coef, const uint16_t* q, uint8_t* out, int stride) {
        constexpr int CB = 13, P1 = 2;
        constexpr int64_t F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299, F1_847 = 15137,
                          F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
        int64_t ws[64];
        for (int c = 0; c < 8; ++c) {
            const int16_t* in = coef + c; const uint16_t* qq = q + c; int64_t* w = ws + c;
            if (!in[8] && !in[16] && !in[24] && !in[32] && !in[40] && !in[48] && !in[56]) {
                const int64_t dcv = (int64_t)in[0] * qq[0] * (1 << P1);
                for (int r = 0; r < 8; ++r) w[8 * r] = dcv;
                continue;
            }
            int64_t z2 = (int64_t)in[16] * qq[16], z3 = (int64_t)in[48] * qq[48];
            int64_t z1 = (z2 + z3) * F0_541;
            int64_t tmp2 = z1 + z3 * (-F1_847), tmp3 = z1 + z2 * F0_765;
            z2 = (int64_t)in[0] * qq[0]; z3 = (int64_t)in[32] * qq[32];
            int64_t tmp0 = (z2 + z3) * (1 << CB), tmp1 = (z2 - z3) * (1 << CB);
            const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = (int64_t)in[56] * qq[56]; tmp1 = (int64_t)in[40] * qq[40]; tmp2 = (int64_t)in[24] * qq[24]; tmp3 = (int64_t)in[8] * qq[8];
            z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; int64_t z4 = tmp1 + tmp3;
            const int64_t z5 = (z3 + z4) * F1_175;
            tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
            z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
            z3 += z5; z4 += z5;
            tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
            w[0] = descale(tmp10 + tmp3, CB - P1); w[56] = descale(tmp10 - tmp3, CB - P1);
            w[8] = descale(tmp11 + tmp2, CB - P1); w[48] = descale(tmp11 - tmp2, CB - P1);
            w[16] = descale(tmp12 + tmp1, CB - P1); w[40] = descale(tmp12 - tmp1, CB - P1);
            w[24] = descale(tmp13 + tmp0, CB - P1); w[32] = descale(tmp13 - tmp0, CB - P1);
        }
        for (int r = 0; r < 8; ++r) {
            const int64_t* w = ws + 8 * r; uint8_t* o = out + (size_t)r * stride;
            int64_t z2 = w[2], z3 = w[6];
            int64_t z1 = (z2 + z3) * F0_541;
            int64_t tmp2 = z1 + z3 * (-F1_847), tmp3 = z1 + z2 * F0_765;
            int64_t tmp0 = (w[0] + w[4]) * (1 << CB), tmp1 = (w[0] - w[4]) * (1 << CB);
            const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
            z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; int64_t z4 = tmp1 + tmp3;
            const int64_t z5 = (z3 + z4) * F1_175;
            tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
            z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
            z3 += z5; z4 += z5;
            tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
            constexpr int SH = CB + P1 + 3;
            o[0] = clamp8(descale(tmp10 + tmp3, SH) + 128); o[7] = clamp8(descale(tmp10 - tmp3, SH) + 128);
            o[1] = clamp8(descale(tmp11 + tmp2, SH) + 128); o[6] = clamp8(descale(tmp11 - tmp2, SH) + 128);
            o[2] = clamp8(descale(tmp12 + tmp1, SH) + 128); o[5] = clamp8(descale(tmp12 - tmp1, SH) + 128);
            o[3] = clamp8(descale(tmp13 + tmp0, SH) + 128); o[4] = clamp8(descale(tmp13 - tmp0, SH) + 128);
        }
    }

    void reconstruct() {
        for (int i = 0; i < ncomp; ++i) {
            Component& c = comp[i];
            if (!qt_defined[c.tq]) throw Corrupt("JPEG: missing quantisation table");
            const int stride = c.blocks_w * 8;
            c.plane.assign((size_t)stride * c.blocks_h * 8, 0);
            for (int by = 0; by < c.blocks_h; ++by)
                for (int bx = 0; bx < c.blocks_w; ++bx)
                    idct_islow(c.coef.data() + ((size_t)by * c.blocks_w + bx) * 64, qt[c.tq], c.plane.data() + (size_t)by * 8 * stride + bx * 8, stride);
            std::vector<int16_t>().swap(c.coef);
        }
    }

    // ---- jdsample.c: one full-resolution row of a component ------------------------------------------------------------
    // row r of the output (0 <= r < height); writes `width` samples (plus up to one spare)
    void upsampled_row(const Component& c, int r, uint8_t* out) const {
        const int stride = c.blocks_w * 8, W = c.width;
        auto src = [&](int y) { y = y < 0 ? 0 : (y >= c.height ? c.height - 1 : y); return c.plane.data() + (size_t)y * stride; }; // jdmainct.c context rows
        if (c.h == hmax && c.v == vmax) { std::memcpy(out, src(r), (size_t)width); return; }
        // jinit_upsampler: the fancy 2:1 horizontal filters are only selected when downsampled_width > 2; narrower components are replicated
        if (c.h * 2 == hmax && W <= 2 && (c.v == vmax || c.v * 2 == vmax)) {
            const uint8_t* in = src(c.v == vmax ? r : (r >> 1));
            for (int x = 0; x < W; ++x) out[2 * x] = out[2 * x + 1] = in[x];
            return;
        }
        if (c.h * 2 == hmax && c.v == vmax) { // h2v1 fancy (pointer walk as jdsample.c)
            const uint8_t* in = src(r);
            uint8_t* o = out;
            int invalue = *in++;
            *o++ = (uint8_t)invalue; *o++ = (uint8_t)((invalue * 3 + in[0] + 2) >> 2);
            for (int colctr = W - 2; colctr > 0; --colctr) {
                invalue = (*in++) * 3;
                *o++ = (uint8_t)((invalue + in[-2] + 1) >> 2);
                *o++ = (uint8_t)((invalue + in[0] + 2) >> 2);
            }
            invalue = *in;
            *o++ = (uint8_t)((invalue * 3 + in[-1] + 1) >> 2); *o++ = (uint8_t)invalue;
            return;
        }
        if (c.h * 2 == hmax && c.v * 2 == vmax) { // h2v2 fancy: nearer row 3/4, farther row 1/4, then the same horizontally
            const int y = r >> 1;
            const uint8_t* in0 = src(y);
            const uint8_t* in1 = src((r & 1) ? y + 1 : y - 1);
            uint8_t* o = out;
            int thiscol = (*in0++) * 3 + (*in1++), nextcol = (*in0++) * 3 + (*in1++), lastcol;
            *o++ = (uint8_t)((thiscol * 4 + 8) >> 4); *o++ = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
            lastcol = thiscol; thiscol = nextcol;
            for (int colctr = W - 2; colctr > 0; --colctr) {
                nextcol = (*in0++) * 3 + (*in1++);
                *o++ = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
                *o++ = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
                lastcol = thiscol; thiscol = nextcol;
            }
            *o++ = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4); *o++ = (uint8_t)((thiscol * 4 + 7) >> 4);
            return;
        }
        if (c.h == hmax && c.v * 2 == vmax) { // h1v2 fancy (libjpeg-turbo jdsample.c h1v2_fancy_upsample)
            const int y = r >> 1;
            const uint8_t* in0 = src(y);
            const uint8_t* in1 = src((r & 1) ? y + 1 : y - 1);
            const int bias = (r & 1) ? 2 : 1;
            for (int x = 0; x < W; ++x) out[x] = (uint8_t)((in0[x] * 3 + in1[x] + bias) >> 2);
            return;
        }
        throw Unsupported("JPEG: this combination of sampling factors is not supported");
    }

    uint8_t* to_rgb() {
        uint8_t* rgb = (uint8_t*)std::malloc((size_t)width * height * 3);
        if (!rgb) throw std::bad_alloc();
        if (ncomp == 1) {
            const int stride = comp[0].blocks_w * 8;
            for (int y = 0; y < height; ++y) {
                const uint8_t* s = comp[0].plane.data() + (size_t)y * stride;
                uint8_t* d = rgb + (size_t)y * width * 3;
                for (int x = 0; x < width; ++x) d[3 * x] = d[3 * x + 1] = d[3 * x + 2] = s[x];
            }
            return rgb;
        }
        bool ycc = true; // jdapimin.c default_decompress_parms
        if (jfif) ycc = true;
        else if (adobe) ycc = adobe_transform != 0;
        else if (comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B') ycc = false;
        // jdcolor.c tables
        int cr_r[256], cb_b[256]; int32_t cr_g[256], cb_g[256];
        for (int i = 0; i < 256; ++i) {
            const int32_t x = i - 128;
            cr_r[i] = (int)((91881 * x + 32768) >> 16);     // FIX(1.40200)
            cb_b[i] = (int)((116130 * x + 32768) >> 16);    // FIX(1.77200)
            cr_g[i] = -46802 * x;                            // FIX(0.71414)
            cb_g[i] = -22554 * x + 32768;                    // FIX(0.34414) + ONE_HALF
        }
        std::vector<uint8_t> rows[3];
        for (int i = 0; i < 3; ++i) rows[i].resize((size_t)width + 16 * hmax + 2);
        try {
            for (int y = 0; y < height; ++y) {
                for (int i = 0; i < 3; ++i) upsampled_row(comp[i], y, rows[i].data());
                uint8_t* d = rgb + (size_t)y * width * 3;
                const uint8_t *Y = rows[0].data(), *Cb = rows[1].data(), *Cr = rows[2].data();
                if (ycc)
                    for (int x = 0; x < width; ++x) {
                        const int yy = Y[x], cb = Cb[x], cr = Cr[x];
                        d[3 * x] = clamp8(yy + cr_r[cr]);
                        d[3 * x + 1] = clamp8(yy + (int)((cb_g[cb] + cr_g[cr]) >> 16));
                        d[3 * x + 2] = clamp8(yy + cb_b[cb]);
                    }
                else
                    for (int x = 0; x < width; ++x) { d[3 * x] = Y[x]; d[3 * x + 1] = Cb[x]; d[3 * x + 2] = Cr[x]; }
            }
        } catch (...) { std::free(rgb); throw; }
        return rgb;
    }

    uint8_t* run() {
        if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) throw Corrupt("JPEG: missing SOI");
        const uint8_t* p = data + 2; const uint8_t* end = data + size;
        bool have_sof = false, have_scan = false;
        while (p + 4 <= end) {
            if (p[0] != 0xFF) { ++p; continue; }
            const int m = p[1];
            if (m == 0xFF) { ++p; continue; }
            if (m == 0x00 || m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { p += 2; continue; }
            if (m == 0xD9) break;
            const int len = be16(p + 2);
            if (len < 2 || p + 2 + len > end) throw Corrupt("JPEG: truncated segment");
            const uint8_t* body = p + 4; const int blen = len - 2;
            if (m == 0xDB) parse_dqt(body, blen);
            else if (m == 0xC4) parse_dht(body, blen);
            else if (m == 0xCC) throw Unsupported("JPEG: arithmetic coding is not supported");
            else if (m >= 0xC0 && m <= 0xCF) { if (have_sof) throw Corrupt("JPEG: two frame headers"); parse_sof(body, blen, m); have_sof = true; }
            else if (m == 0xDD) { if (blen >= 2) restart_interval = be16(body); }
            else if (m == 0xE0) { if (blen >= 5 && !std::memcmp(body, "JFIF", 5)) jfif = true; }
            else if (m == 0xEE) { if (blen >= 12 && !std::memcmp(body, "Adobe", 5)) { adobe = true; adobe_transform = body[11]; } }
            else if (m == 0xDA) {
                if (!have_sof) throw Corrupt("JPEG: scan before the frame header");
                p = decode_scan(body, blen);
                have_scan = true;
                continue;
            }
            p += 2 + len;
        }
        if (!have_sof || !have_scan) throw Corrupt("JPEG: no image data");
        if (adobe && adobe_transform == 2) throw Unsupported("JPEG: YCCK is not supported");
        reconstruct();
        return to_rgb();
    }
};

} // namespace lfs_jpeg

// entry used by lfs_io.cpp: returns a malloc'ed [h,w,3] buffer, or throws (Unsupported -> LFS_IO_E_UNSUPPORTED, anything else -> LFS_IO_E_FORMAT)
uint8_t* lfs_decode_jpeg_rgb8(const uint8_t* data, size_t size, int32_t* width, int32_t* height, bool* unsupported, std::string* error) {
    *unsupported = false;
    try {
        lfs_jpeg::Decoder d(data, size);
        uint8_t* out = d.run();
        *width = d.width; *height = d.height;
        return out;
    } catch (const lfs_jpeg::Unsupported& e) {
        *unsupported = true; *error = e.what();
    } catch (const std::exception& e) {
        *error = e.what();
    }
    return nullptr;
}
