// usc_host_objects.cpp - the one HOST stage of the per-frame path: remove_invalid_objects
// (reference unscreen/utils/maskprocess.py:77-152, score map :155-189).
//
// The reference finds objects with cv2.findContours(RETR_LIST) - sequential border following - between the
// (device) difference-mask stage and the (device) trimap stage, so this stage runs on the CPU here too, on a
// 1 byte/pixel mask the driver copies down and back up.  Structure, chosen for a per-frame budget of ~1 ms:
//   * one raster scan of a zero-padded 0/1 image, 8 bytes at a time over uniform runs;
//   * each border is followed once (Suzuki-Abe, the rule set of OpenCV's tracer) while the shoelace area,
//     the bounding rows and the vertical crossings are accumulated on the fly - no point list is kept;
//   * the filled region (polygon pixels + even-odd interior = what drawContours(FILLED) paints) is walked as
//     per-row spans from the sorted crossings, with a per-object stamp image so every pixel counts once;
//   * the float64 score map is cached by the caller per geometry instead of being rebuilt per call.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/unscreen_hip.h"

namespace {

// numpy.linspace(a, b, n)[i]
inline double lin(double a, double b, int n, int i) {
    if (n == 1) return a;
    if (i == n - 1) return b;
    return (double)i * ((b - a) / (double)(n - 1)) + a;
}

// maskprocess.py:155-178: distance-to-centre ramps, sqrt, normalised and inverted.
void build_score(int h, int w, double cy, double cx, double* out) {
    const int y0 = (int)(h * cy), x0 = (int)(w * cx);
    std::vector<double> col(w), row(h);
    for (int x = 0; x < w; x++) { double v = x >= x0 ? lin(0, 1, w - x0, x - x0) : lin(1, 0, x0, x); col[x] = v * v; }
    for (int y = 0; y < h; y++) { double v = y >= y0 ? lin(0, 1, h - y0, y - y0) : lin(1, 0, y0, y); row[y] = v * v; }
    double mx = -1;
    for (int y = 0; y < h; y++) {
        double* o = out + (size_t)y * w;
        const double r = row[y];
        for (int x = 0; x < w; x++) { double s = sqrt(col[x] + r); o[x] = s; if (s > mx) mx = s; }
    }
    const size_t n = (size_t)h * w;
    for (size_t i = 0; i < n; i++) out[i] = (mx - out[i]) / mx;
}

struct Border {
    double twice_area = 0;                 // signed shoelace sum
    std::vector<uint32_t> cross;           // (row << 16 | col) of every vertical crossing (upper end of the move)
    std::vector<uint32_t> pix;             // (row << 16 | col) of every border pixel visit
};

const int DX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
const int DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};

// Follow one border starting at padded position (x, y).  Marks: 2 = visited, -126 = visited and the pixel
// to the east was found empty (so the raster scan will not start a hole there).
void follow(int8_t* img, int step, int x, int y, bool hole, Border& b) {
    int off[16];
    for (int k = 0; k < 8; k++) off[k] = off[k + 8] = DY[k] * step + DX[k];
    b.twice_area = 0; b.cross.clear(); b.pix.clear();
    int8_t* const p0 = img + (size_t)y * step + x;
    int s_end = hole ? 0 : 4, s = s_end;
    int8_t* p1;
    do { s = (s - 1) & 7; p1 = p0 + off[s]; } while (*p1 == 0 && s != s_end);
    if (s == s_end) {                      // single pixel
        *p0 = (int8_t)(2 | -128);
        b.pix.push_back((uint32_t)(y - 1) << 16 | (uint32_t)(x - 1));
        return;
    }
    int8_t* cur = p0;
    int cx = x - 1, cy = y - 1;            // un-padded coordinates
    for (;;) {
        s_end = s;
        int8_t* nxt;
        for (;;) { nxt = cur + off[++s]; if (*nxt != 0) break; }
        s &= 7;
        if ((unsigned)(s - 1) < (unsigned)s_end) *cur = (int8_t)(2 | -128);
        else if (*cur == 1) *cur = 2;
        b.pix.push_back((uint32_t)cy << 16 | (uint32_t)cx);
        const int nx = cx + DX[s], ny = cy + DY[s];
        b.twice_area += (double)cx * ny - (double)cy * nx;
        if (ny != cy) {
            if (ny > cy) b.cross.push_back((uint32_t)cy << 16 | (uint32_t)cx);
            else b.cross.push_back((uint32_t)ny << 16 | (uint32_t)nx);
        }
        cx = nx; cy = ny;
        if (nxt == p0 && cur == p1) break;
        cur = nxt;
        s = (s + 4) & 7;
    }
}

// The filled border = its pixels + the even-odd interior spans [xa, xb) between sorted crossing pairs of a row.  Spans of
// one contour are pairwise disjoint, so only the border pixels need the stamp image (a border can visit a pixel twice,
// and border pixels may lie inside a span).
struct FillStats { double ssum, csum; size_t cnt; };
FillStats filled_stats(Border& b, uint32_t* __restrict stamp, uint32_t id, int W, const double* __restrict score, const uint8_t* __restrict seg) {
    FillStats st = {0., 0., 0};
    for (uint32_t p : b.pix) {
        const size_t o = (size_t)(p >> 16) * W + (p & 0xffff);
        if (stamp[o] != id) { stamp[o] = id; st.ssum += score[o]; st.csum += seg[o]; st.cnt++; }
    }
    std::sort(b.cross.begin(), b.cross.end());
    const size_t nc = b.cross.size();
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    uint64_t c = 0, k = 0;
    for (size_t i = 0; i + 1 < nc; i += 2) {              // crossings pair up within a row (closed curve)
        const uint32_t a = b.cross[i], e = b.cross[i + 1];
        const size_t base = (size_t)(a >> 16) * W;
        const int xa = (int)(a & 0xffff), xb = (int)(e & 0xffff);
        const uint32_t* __restrict sr = stamp + base;
        const double* __restrict sc = score + base;
        const uint8_t* __restrict sg = seg + base;
        int x = xa;
        for (; x + 4 <= xb; x += 4) {
            const bool n0 = sr[x] != id, n1 = sr[x + 1] != id, n2 = sr[x + 2] != id, n3 = sr[x + 3] != id;
            s0 += n0 ? sc[x] : 0.; s1 += n1 ? sc[x + 1] : 0.; s2 += n2 ? sc[x + 2] : 0.; s3 += n3 ? sc[x + 3] : 0.;
            c += (n0 ? sg[x] : 0) + (n1 ? sg[x + 1] : 0) + (n2 ? sg[x + 2] : 0) + (n3 ? sg[x + 3] : 0);
            k += (int)n0 + (int)n1 + (int)n2 + (int)n3;
        }
        for (; x < xb; x++)
            if (sr[x] != id) { s0 += sc[x]; c += sg[x]; k++; }
    }
    st.ssum += (s0 + s1) + (s2 + s3);
    st.csum += (double)c;                                  // integer sum: exact
    st.cnt += k;
    return st;
}

void mark_filled(const Border& b, uint8_t* __restrict valid, int W) {        // b.cross is already sorted
    const size_t nc = b.cross.size();
    for (size_t i = 0; i + 1 < nc; i += 2) {
        const uint32_t a = b.cross[i], e = b.cross[i + 1];
        const int xa = (int)(a & 0xffff), xb = (int)(e & 0xffff);
        if (xb > xa) memset(valid + (size_t)(a >> 16) * W + xa, 1, (size_t)(xb - xa));
    }
    for (uint32_t p : b.pix) valid[(size_t)(p >> 16) * W + (p & 0xffff)] = 1;
}

}  // namespace

extern "C" int usc_remove_invalid_objects_host(uint8_t* alpha, const uint8_t* seg, int H, int W,
                                               double center_y, double center_x, double saliency_thr, double consensus_thr,
                                               double* score_cache, int* score_cache_valid) {
    if (!alpha || H <= 0 || W <= 0 || H > 65535 || W > 65535) return USC_E_ARG;
    const size_t n = (size_t)H * W;
    std::vector<double> own_score;
    double* score = score_cache;
    if (!score) { own_score.resize(n); score = own_score.data(); }
    if (!score_cache || !score_cache_valid || !*score_cache_valid) {
        build_score(H, W, center_y, center_x, score);
        if (score_cache && score_cache_valid) *score_cache_valid = 1;
    }
    // Scratch that survives between calls on this thread: fresh 8 MB vectors cost more in page faults and zero fills than
    // the tracing itself.  `stamp` is never cleared - ids keep counting up across calls (wrap-around: one clear).
    struct Scratch { std::vector<int8_t> img; std::vector<uint32_t> stamp; std::vector<uint8_t> valid; uint32_t next_id = 0; };
    static thread_local Scratch sc;
    if (!seg) seg = alpha;              // consensus against the input itself: alpha is only written after the last object

    const int step = W + 2;
    const size_t padded = (size_t)(H + 2) * step;
    if (sc.img.size() != padded) sc.img.assign(padded, 0);            // the border stays zero; the interior is rewritten below
    for (int y = 0; y < H; y++) {
        const uint8_t* __restrict a = alpha + (size_t)y * W;
        int8_t* __restrict d = sc.img.data() + (size_t)(y + 1) * step + 1;
        for (int x = 0; x < W; x++) d[x] = (int8_t)(a[x] != 0);
    }
    if (sc.stamp.size() != n || sc.next_id > 0xf0000000u) { sc.stamp.assign(n, 0); sc.next_id = 0; }
    sc.valid.assign(n, 0);
    std::vector<int8_t>& img = sc.img;
    std::vector<uint32_t>& stamp = sc.stamp;
    std::vector<uint8_t>& valid = sc.valid;
    static thread_local Border b;
    uint32_t id = sc.next_id;
    const double denom = (double)H * (double)W;
    for (int y = 1; y <= H; y++) {
        int8_t* row = img.data() + (size_t)y * step;
        int prev = 0;
        for (int x = 1; x <= W; x++) {
            int p = row[x];
            if (p == prev) {
                // fast-forward over 8-byte runs equal to the current value
                uint64_t pat = 0x0101010101010101ull * (uint8_t)p;
                while (x + 8 <= W) {
                    uint64_t q;
                    memcpy(&q, row + x + 1, 8);
                    if (q != pat) break;
                    x += 8;
                }
                continue;
            }
            bool hole = false, start = false;
            if (prev == 0 && p == 1) start = true;
            else if (p == 0 && prev >= 1) { hole = true; start = true; }
            if (start) {
                follow(img.data(), step, x - (hole ? 1 : 0), y, hole, b);
                ++id;
                if (fabs(b.twice_area * 0.5) >= 100) {
                    const FillStats fs = filled_stats(b, stamp.data(), id, W, score, seg);
                    const double sal = fs.ssum / denom;
                    const double cons = (fs.csum / (double)fs.cnt) / 255.;
                    if ((sal > saliency_thr && cons > consensus_thr) || sal > saliency_thr * 10) mark_filled(b, valid.data(), W);
                }
                p = row[x];
            }
            prev = p;
        }
    }
    sc.next_id = id;
    {
        uint8_t* __restrict ap = alpha;
        const uint8_t* __restrict vp = valid.data();
        for (size_t i = 0; i < n; i++) ap[i] &= (uint8_t)(0 - (vp[i] != 0));
    }
    return USC_OK;
}
