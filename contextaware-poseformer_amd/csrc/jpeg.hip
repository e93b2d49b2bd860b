// N3, first half: the per-sample image decode of Human36M.__getitem__ (datasets/human36m.py:292-295: cv2.imread(path, IMREAD_COLOR |
// IMREAD_IGNORE_ORIENTATION) -> uint8 BGR [H][W][3]) for baseline JPEG, in front of capf_warp_affine (the crop, :297-300).
//
// Split the way the data allows: the entropy-coded segment is a serial bit stream -- the HOST walks it (Huffman decode, DC prediction,
// restart markers) into quantised coefficient blocks; everything after that is independent per 8x8 block / per pixel and runs on the
// GPU: dequantisation + inverse DCT, chroma upsampling, YCbCr -> BGR.  cv2.imread and Pillow both decode through libjpeg(-turbo) with its
// defaults, and those defaults are integer algorithms with published definitions; this file restates them so that the result is
// BIT-EXACT against such a decode (tests/test_jpeg.py holds it to Pillow's libjpeg-turbo on 4:4:4 / 4:2:2 / 4:2:0 / grey images):
//   * inverse DCT: jidctint.c "islow" (JDCT_ISLOW, the default): 13-bit constants, two passes, 2 extra bits between them, range limit;
//   * upsampling: jdsample.c "fancy" (do_fancy_upsampling = TRUE, the default): triangle filter, h2v1 (3/4, 1/4 with alternating
//     rounding) and h2v2 (9/16, 3/16, 3/16, 1/16 with +8 / +7), edge rows / columns replicated;
//   * colour: jdcolor.c ycc_rgb_convert: 16-bit fixed point tables, R = y + 1.40200 cr, G = y - 0.34414 cb - 0.71414 cr, B = y + 1.77200 cb.
// Supported: baseline / extended sequential Huffman (SOF0 / SOF1), 8 bit, 1 or 3 components in ONE interleaved scan, sampling
// 1x1 (4:4:4), 2x1 (4:2:2), 2x2 (4:2:0) with 1x1 chroma, restart intervals.  Progressive / arithmetic / CMYK: CAPF_ERR_UNSUPPORTED.
#include <string.h>

#include <vector>

#include "capf.h"
#include "kernels.h"

namespace capf {

struct JpegComp { int id, h, v, tq, td, ta; int bw, bh; size_t coef_off; int pw, ph; size_t plane_off; };   // bw / bh: blocks; pw / ph: plane size in samples
struct JpegHeader {
    int W = 0, H = 0, nc = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0, restart = 0;
    JpegComp c[3];
    unsigned short qt[4][64];
    bool have_qt[4] = {false, false, false, false};
    unsigned char bits[2][4][17], vals[2][4][256];
    bool have_ht[2][4] = {{false, false, false, false}, {false, false, false, false}};
    size_t scan_off = 0;              // first byte of the entropy-coded segment
    size_t coef_elems = 0, plane_bytes = 0;
};

static const unsigned char kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                          35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// marker segments up to and including SOS; CAPF_OK / CAPF_ERR_INVALID (not a JPEG, truncated) / CAPF_ERR_UNSUPPORTED (a JPEG this path does not decode)
static int jpeg_parse(const unsigned char* d, size_t n, JpegHeader& h) {
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return CAPF_ERR_INVALID;
    size_t p = 2;
    bool have_sof = false;
    while (p + 4 <= n) {
        if (d[p] != 0xFF) return CAPF_ERR_INVALID;
        while (p < n && d[p] == 0xFF) ++p;            // fill bytes
        if (p >= n) return CAPF_ERR_INVALID;
        const int m = d[p++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) return CAPF_ERR_INVALID;
        if (p + 2 > n) return CAPF_ERR_INVALID;
        const size_t len = ((size_t)d[p] << 8) | d[p + 1];
        if (len < 2 || p + len > n) return CAPF_ERR_INVALID;
        const unsigned char* s = d + p + 2;
        const size_t sl = len - 2;
        if (m == 0xC0 || m == 0xC1) {
            if (sl < 6 || s[0] != 8) return CAPF_ERR_UNSUPPORTED;
            h.H = (s[1] << 8) | s[2]; h.W = (s[3] << 8) | s[4]; h.nc = s[5];
            if (h.W <= 0 || h.H <= 0 || (h.nc != 1 && h.nc != 3) || sl < (size_t)(6 + 3 * h.nc)) return CAPF_ERR_UNSUPPORTED;
            for (int i = 0; i < h.nc; ++i) {
                h.c[i].id = s[6 + 3 * i]; h.c[i].h = s[7 + 3 * i] >> 4; h.c[i].v = s[7 + 3 * i] & 15; h.c[i].tq = s[8 + 3 * i] & 3;
            }
            have_sof = true;
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            return CAPF_ERR_UNSUPPORTED;              // progressive, lossless, arithmetic, hierarchical
        } else if (m == 0xDB) {
            size_t q = 0;
            while (q < sl) {
                const int pq = s[q] >> 4, tq = s[q] & 15;
                if (tq > 3 || q + 1 + (pq ? 128 : 64) > sl) return CAPF_ERR_INVALID;
                for (int i = 0; i < 64; ++i) h.qt[tq][kZigzag[i]] = pq ? (unsigned short)((s[q + 1 + 2 * i] << 8) | s[q + 2 + 2 * i]) : s[q + 1 + i];
                h.have_qt[tq] = true;
                q += 1 + (pq ? 128 : 64);
            }
        } else if (m == 0xC4) {
            size_t q = 0;
            while (q < sl) {
                const int tc = s[q] >> 4, th = s[q] & 15;
                if (tc > 1 || th > 3 || q + 17 > sl) return CAPF_ERR_INVALID;
                int cnt = 0;
                h.bits[tc][th][0] = 0;
                for (int i = 1; i <= 16; ++i) { h.bits[tc][th][i] = s[q + i]; cnt += s[q + i]; }
                if (cnt > 256 || q + 17 + cnt > sl) return CAPF_ERR_INVALID;
                // the counts must form a prefix code (jdhuff.c jpeg_make_d_derived_tbl): after the codes of length l the next code may not
                // exceed 2^l -- an oversubscribed table would index past the decoder's lookup arrays
                for (int l = 1, code = 0; l <= 16; ++l) {
                    code += h.bits[tc][th][l];
                    if (code > (1 << l)) return CAPF_ERR_INVALID;
                    code <<= 1;
                }
                // symbols: a DC symbol is a magnitude category (<= 11 for 8-bit baseline; <= 15 checked here, the decoder refuses > 11),
                // an AC symbol's low nibble likewise (<= 10)
                for (int i = 0; i < cnt; ++i)
                    if (tc == 0 ? s[q + 17 + i] > 15 : (s[q + 17 + i] & 15) > 10) return CAPF_ERR_INVALID;
                memcpy(h.vals[tc][th], s + q + 17, cnt);
                h.have_ht[tc][th] = true;
                q += 17 + cnt;
            }
        } else if (m == 0xDD) {
            if (sl < 2) return CAPF_ERR_INVALID;
            h.restart = (s[0] << 8) | s[1];
        } else if (m == 0xDA) {
            if (!have_sof || sl < 1 || s[0] != h.nc || sl < (size_t)(1 + 2 * h.nc + 3)) return CAPF_ERR_UNSUPPORTED;      // one interleaved scan
            for (int i = 0; i < h.nc; ++i) {
                if (s[1 + 2 * i] != h.c[i].id) return CAPF_ERR_UNSUPPORTED;
                h.c[i].td = s[2 + 2 * i] >> 4; h.c[i].ta = s[2 + 2 * i] & 15;
                if (h.c[i].td > 3 || h.c[i].ta > 3 || !h.have_ht[0][h.c[i].td] || !h.have_ht[1][h.c[i].ta] || !h.have_qt[h.c[i].tq]) return CAPF_ERR_INVALID;
            }
            h.scan_off = p + len;
            break;
        }
        p += len;
    }
    if (!have_sof || !h.scan_off) return CAPF_ERR_INVALID;
    if (h.nc == 1) { h.c[0].h = h.c[0].v = 1; }
    else {
        if (h.c[1].h != 1 || h.c[1].v != 1 || h.c[2].h != 1 || h.c[2].v != 1) return CAPF_ERR_UNSUPPORTED;
        if (!((h.c[0].h == 1 && h.c[0].v == 1) || (h.c[0].h == 2 && h.c[0].v == 1) || (h.c[0].h == 2 && h.c[0].v == 2))) return CAPF_ERR_UNSUPPORTED;
    }
    h.hmax = h.c[0].h; h.vmax = h.c[0].v;
    h.mcux = (h.W + 8 * h.hmax - 1) / (8 * h.hmax);
    h.mcuy = (h.H + 8 * h.vmax - 1) / (8 * h.vmax);
    size_t co = 0, po = 0;
    for (int i = 0; i < h.nc; ++i) {
        JpegComp& c = h.c[i];
        c.bw = h.mcux * c.h; c.bh = h.mcuy * c.v;
        c.coef_off = co; co += (size_t)c.bw * c.bh * 64;
        c.pw = c.bw * 8; c.ph = c.bh * 8;
        c.plane_off = po; po += (size_t)c.pw * c.ph;
    }
    h.coef_elems = co;
    h.plane_bytes = (po + 15) & ~(size_t)15;
    return CAPF_OK;
}

// ---- host entropy decoder (ITU T.81 F.2.2): canonical Huffman codes through a 9-bit lookup + the slow path ----
struct HuffTab {
    int maxcode[18], valptr[17];
    unsigned char look_n[512], look_v[512];
    const unsigned char* vals;
    void build(const unsigned char* bits, const unsigned char* v) {
        vals = v;
        int code = 0, k = 0;
        unsigned short codes[256];
        unsigned char sizes[256];
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k - code;                      // vals index = valptr[l] + code for codes of length l
            for (int i = 0; i < bits[l]; ++i) { codes[k] = (unsigned short)code; sizes[k] = (unsigned char)l; ++k; ++code; }
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7FFFFFFF;
        memset(look_n, 0, sizeof(look_n));
        for (int i = 0; i < k; ++i)
            if (sizes[i] <= 9) {
                const int base = codes[i] << (9 - sizes[i]);
                for (int j = 0; j < (1 << (9 - sizes[i])); ++j) { look_n[base + j] = sizes[i]; look_v[base + j] = v[i]; }
            }
    }
};

struct BitReader {
    const unsigned char* d;
    size_t n, p;
    unsigned long long acc = 0;
    int cnt = 0;
    bool hit_marker = false;
    void fill() {
        while (cnt <= 48) {
            unsigned b = 0;
            if (!hit_marker && p < n) {
                b = d[p];
                if (b == 0xFF) {
                    if (p + 1 < n && d[p + 1] == 0x00) p += 2;
                    else { hit_marker = true; b = 0; }   // a marker: feed zeros until the caller deals with it
                } else ++p;
            }
            acc = (acc << 8) | b;
            cnt += 8;
        }
    }
    int peek(int k) { if (cnt < k) fill(); return (int)((acc >> (cnt - k)) & ((1ull << k) - 1)); }
    void skip(int k) { cnt -= k; }
    int get(int k) { if (k == 0) return 0; const int v = peek(k); skip(k); return v; }
    void align_and_restart() {                        // at a restart boundary: drop the partial byte, step over RSTn
        acc = 0; cnt = 0;
        if (hit_marker) { hit_marker = false; }
        while (p + 1 < n && !(d[p] == 0xFF && d[p + 1] >= 0xD0 && d[p + 1] <= 0xD7)) ++p;
        if (p + 1 < n) p += 2;
    }
};

static inline int huff_decode(BitReader& br, const HuffTab& t) {
    const int look = br.peek(9);
    if (t.look_n[look]) { br.skip(t.look_n[look]); return t.look_v[look]; }
    int code = br.peek(16), l = 10;
    for (; l <= 16; ++l) {
        const int c = code >> (16 - l);
        if (c <= t.maxcode[l]) { br.skip(l); return t.vals[t.valptr[l] + c]; }
    }
    br.skip(16);
    return 0;                                         // corrupt stream: keep going, like libjpeg's warning path
}
static inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

// coefficients in natural (row-major) order, component planes of blocks [bh][bw][64]
static int jpeg_entropy_decode(const unsigned char* d, size_t n, const JpegHeader& h, short* coef) {
    memset(coef, 0, h.coef_elems * sizeof(short));
    HuffTab dc[4], ac[4];
    for (int t = 0; t < 4; ++t) {
        if (h.have_ht[0][t]) dc[t].build(h.bits[0][t], h.vals[0][t]);
        if (h.have_ht[1][t]) ac[t].build(h.bits[1][t], h.vals[1][t]);
    }
    BitReader br{d, n, h.scan_off};
    int pred[3] = {0, 0, 0};
    int until_restart = h.restart;
    for (int my = 0; my < h.mcuy; ++my)
        for (int mx = 0; mx < h.mcux; ++mx) {
            if (h.restart && until_restart == 0) {
                br.align_and_restart();
                pred[0] = pred[1] = pred[2] = 0;
                until_restart = h.restart;
            }
            for (int ci = 0; ci < h.nc; ++ci) {
                const JpegComp& c = h.c[ci];
                for (int by = 0; by < c.v; ++by)
                    for (int bx = 0; bx < c.h; ++bx) {
                        short* blk = coef + c.coef_off + ((size_t)(my * c.v + by) * c.bw + (mx * c.h + bx)) * 64;
                        int s = huff_decode(br, dc[c.td]);
                        if (s > 11) return CAPF_ERR_INVALID;       // (8-bit baseline: DC differences have at most 11 magnitude bits)
                        if (s) { const int r = br.get(s); pred[ci] += extend(r, s); }
                        blk[0] = (short)pred[ci];
                        for (int k = 1; k < 64;) {
                            const int rs = huff_decode(br, ac[c.ta]);
                            const int r = rs >> 4, sz = rs & 15;
                            if (sz == 0) {
                                if (r != 15) break;    // EOB
                                k += 16;
                                continue;
                            }
                            k += r;
                            if (k > 63) break;
                            if (sz > 10) return CAPF_ERR_INVALID;
                            blk[kZigzag[k]] = (short)extend(br.get(sz), sz);
                            ++k;
                        }
                    }
            }
            --until_restart;
        }
    return CAPF_OK;
}

// ---- GPU half -------------------------------------------------------------------------------------------------------------------
struct JpegDev {
    int W, H, nc, hmax, vmax;
    int bw[3], bh[3], pw[3], ph[3], dw[3], dh[3];      // blocks, plane size, true ("downsampled") size of each component
    long coef_off[3], plane_off[3];
    unsigned short qt[3][64];
};

// jidctint.c jpeg_idct_islow on one dequantised block per thread; samples into the component's plane
__global__ __launch_bounds__(64) void jpeg_idct_kernel(const short* __restrict__ coef, unsigned char* __restrict__ planes, JpegDev jd, int comp, int nblocks) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= nblocks) return;
    constexpr int CB = 13, P1 = 2;
    constexpr long F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137, F1961 = 16069,
                   F2053 = 16819, F2562 = 20995, F3072 = 25172;
    const short* in = coef + jd.coef_off[comp] + (long)b * 64;
    const unsigned short* q = jd.qt[comp];
    long ws[64];
    auto pass = [&](const long v[8], long o[8], int shift) {
        long z2 = v[2], z3 = v[6];
        long z1 = (z2 + z3) * F0541;
        long tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
        z2 = v[0]; z3 = v[4];
        long tmp0 = (z2 + z3) << CB, tmp1 = (z2 - z3) << CB;
        const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = v[7]; tmp1 = v[5]; tmp2 = v[3]; tmp3 = v[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        long z4 = tmp1 + tmp3;
        const long z5 = (z3 + z4) * F1175;
        tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
        z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        const long r = 1L << (shift - 1);
        o[0] = (tmp10 + tmp3 + r) >> shift; o[7] = (tmp10 - tmp3 + r) >> shift;
        o[1] = (tmp11 + tmp2 + r) >> shift; o[6] = (tmp11 - tmp2 + r) >> shift;
        o[2] = (tmp12 + tmp1 + r) >> shift; o[5] = (tmp12 - tmp1 + r) >> shift;
        o[3] = (tmp13 + tmp0 + r) >> shift; o[4] = (tmp13 - tmp0 + r) >> shift;
    };
    for (int c = 0; c < 8; ++c) {                      // pass 1: columns (the all-zero-AC shortcut of the C code gives the same numbers)
        long v[8], o[8];
        for (int r = 0; r < 8; ++r) v[r] = (long)in[r * 8 + c] * (long)q[r * 8 + c];
        pass(v, o, CB - P1);
        for (int r = 0; r < 8; ++r) ws[r * 8 + c] = o[r];
    }
    const int by = b / jd.bw[comp], bx = b - by * jd.bw[comp];
    unsigned char* dst = planes + jd.plane_off[comp] + (long)(by * 8) * jd.pw[comp] + bx * 8;
    for (int r = 0; r < 8; ++r) {                      // pass 2: rows, + 128, clamp
        long o[8];
        pass(ws + r * 8, o, CB + P1 + 3);
        unsigned long long pk = 0;
        for (int c = 0; c < 8; ++c) {
            // range_limit[x & RANGE_MASK] of jdmaster.c prepare_range_limit_table: the 10-bit masked index selects, in order, [0, 127] -> x + 128,
            // [128, 511] -> 255, [512, 895] -> 0, [896, 1023] -> x - 1024 + 128 (i.e. the clamp of x + 128 for every x in [-512, 511])
            const int x = (int)(o[c] & 1023);
            const int sv = x < 128 ? x + 128 : (x < 512 ? 255 : (x < 896 ? 0 : x - 896));
            pk |= (unsigned long long)sv << (8 * c);
        }
        *reinterpret_cast<unsigned long long*>(dst + (long)r * jd.pw[comp]) = pk;
    }
}

// chroma sample at full resolution: jdsample.c fullsize / h2v1_fancy / h2v2_fancy with libjpeg's edge handling (context rows replicated at the
// top and below the component's last true row; first / last column special-cased)
__device__ __forceinline__ int jpeg_chroma(const unsigned char* pl, int pw, int dw, int dh, int hs, int vs, int x, int y) {
    if (hs == 1 && vs == 1) return pl[(long)y * pw + x];
    const int cx = x >> 1, odd = x & 1;
    if (vs == 1) {                                      // h2v1
        const unsigned char* r = pl + (long)y * pw;
        const int v = r[cx];
        if (dw == 1) return v;
        if (!odd) return cx == 0 ? v : (v * 3 + r[cx - 1] + 1) >> 2;
        return cx == dw - 1 ? v : (v * 3 + r[cx + 1] + 2) >> 2;
    }
    const int cy = y >> 1;                              // h2v2: nearer row cy, farther row cy -/+ 1
    const int fy = min(max((y & 1) ? cy + 1 : cy - 1, 0), dh - 1);
    const unsigned char* r0 = pl + (long)cy * pw;
    const unsigned char* r1 = pl + (long)fy * pw;
    const int cur = r0[cx] * 3 + r1[cx];
    if (!odd) {
        if (cx == 0) return (cur * 4 + 8) >> 4;
        return (cur * 3 + (r0[cx - 1] * 3 + r1[cx - 1]) + 8) >> 4;
    }
    if (cx == dw - 1) return (cur * 4 + 7) >> 4;
    return (cur * 3 + (r0[cx + 1] * 3 + r1[cx + 1]) + 7) >> 4;
}

__global__ __launch_bounds__(256) void jpeg_color_kernel(const unsigned char* __restrict__ planes, unsigned char* __restrict__ out, JpegDev jd, long out_pitch) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= jd.W || y >= jd.H) return;
    const int Y = planes[jd.plane_off[0] + (long)y * jd.pw[0] + x];
    int r = Y, g = Y, b = Y;
    if (jd.nc == 3) {
        const int cb = jpeg_chroma(planes + jd.plane_off[1], jd.pw[1], jd.dw[1], jd.dh[1], jd.hmax, jd.vmax, x, y) - 128;
        const int cr = jpeg_chroma(planes + jd.plane_off[2], jd.pw[2], jd.dw[2], jd.dh[2], jd.hmax, jd.vmax, x, y) - 128;
        // jdcolor.c build_ycc_rgb_table, SCALEBITS 16: FIX(1.40200) = 91881, FIX(1.77200) = 116130, FIX(0.71414) = 46802, FIX(0.34414) = 22554
        r = Y + ((91881 * cr + 32768) >> 16);
        b = Y + ((116130 * cb + 32768) >> 16);
        g = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
        r = min(max(r, 0), 255); g = min(max(g, 0), 255); b = min(max(b, 0), 255);
    }
    unsigned char* o = out + (long)y * out_pitch + 3L * x;
    o[0] = (unsigned char)b; o[1] = (unsigned char)g; o[2] = (unsigned char)r;
}

}  // namespace capf

using capf::JpegHeader;

int capf_jpeg_info(const uint8_t* data, size_t n, int32_t* width, int32_t* height, int32_t* components, int32_t* h_samp, int32_t* v_samp,
                   size_t* scratch_bytes) {
    if (!data) return CAPF_ERR_INVALID;
    JpegHeader h;
    const int rc = capf::jpeg_parse(data, n, h);
    if (rc != CAPF_OK) return rc;
    if (width) *width = h.W;
    if (height) *height = h.H;
    if (components) *components = h.nc;
    if (h_samp) *h_samp = h.hmax;
    if (v_samp) *v_samp = h.vmax;
    if (scratch_bytes) *scratch_bytes = ((h.coef_elems * sizeof(short) + 15) & ~(size_t)15) + h.plane_bytes;
    return CAPF_OK;
}

// host half only: the quantised coefficients, natural order, component after component, blocks [bh][bw][64] (tests; no GPU)
int capf_jpeg_coefficients(const uint8_t* data, size_t n, int16_t* coef, size_t coef_elems) {
    if (!data || !coef) return CAPF_ERR_INVALID;
    JpegHeader h;
    const int rc = capf::jpeg_parse(data, n, h);
    if (rc != CAPF_OK) return rc;
    if (coef_elems < h.coef_elems) return CAPF_ERR_INVALID;
    return capf::jpeg_entropy_decode(data, n, h, coef);
}

int capf_jpeg_decode(void* stream, const uint8_t* data, size_t n, uint8_t* out_bgr, size_t out_pitch_bytes, void* scratch, size_t scratch_bytes) {
    if (!data || !out_bgr || !scratch) return CAPF_ERR_INVALID;
    JpegHeader h;
    int rc = capf::jpeg_parse(data, n, h);
    if (rc != CAPF_OK) return rc;
    const size_t coef_bytes = (h.coef_elems * sizeof(short) + 15) & ~(size_t)15;
    if (scratch_bytes < coef_bytes + h.plane_bytes || out_pitch_bytes < (size_t)h.W * 3) return CAPF_ERR_INVALID;
    static thread_local std::vector<short> host;      // (one decode at a time per thread: the upload below is waited for before the next reuse)
    host.resize(h.coef_elems);
    rc = capf::jpeg_entropy_decode(data, n, h, host.data());
    if (rc != CAPF_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    short* d_coef = static_cast<short*>(scratch);
    unsigned char* d_planes = static_cast<unsigned char*>(scratch) + coef_bytes;
    if (hipMemcpyAsync(d_coef, host.data(), h.coef_elems * sizeof(short), hipMemcpyHostToDevice, s) != hipSuccess) return CAPF_ERR_HIP;
    capf::JpegDev jd{};
    jd.W = h.W; jd.H = h.H; jd.nc = h.nc; jd.hmax = h.hmax; jd.vmax = h.vmax;
    for (int i = 0; i < h.nc; ++i) {
        const capf::JpegComp& c = h.c[i];
        jd.bw[i] = c.bw; jd.bh[i] = c.bh; jd.pw[i] = c.pw; jd.ph[i] = c.ph;
        jd.dw[i] = (h.W * c.h + h.hmax - 1) / h.hmax; jd.dh[i] = (h.H * c.v + h.vmax - 1) / h.vmax;
        jd.coef_off[i] = (long)c.coef_off; jd.plane_off[i] = (long)c.plane_off;
        memcpy(jd.qt[i], h.qt[c.tq], sizeof(jd.qt[i]));
    }
    for (int i = 0; i < h.nc; ++i) {
        const int nb = jd.bw[i] * jd.bh[i];
        hipLaunchKernelGGL(capf::jpeg_idct_kernel, dim3((nb + 63) / 64), dim3(64), 0, s, d_coef, d_planes, jd, i, nb);
    }
    hipLaunchKernelGGL(capf::jpeg_color_kernel, dim3((h.W + 63) / 64, (h.H + 3) / 4), dim3(256), 0, s, d_planes, out_bgr, jd, (long)out_pitch_bytes);
    if (hipGetLastError() != hipSuccess) return CAPF_ERR_HIP;
    // the host staging buffer is reused by this thread's next decode: wait for the upload (a loader-side call, not part of capf_forward)
    return hipStreamSynchronize(s) == hipSuccess ? CAPF_OK : CAPF_ERR_HIP;
}
