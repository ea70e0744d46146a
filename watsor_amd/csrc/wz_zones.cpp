// Host-side zone extraction from a mask's alpha plane, without OpenCV.
//
// Stands in for `MaskFilter.__init__` + `find_contours` + `contours_key`
// (`watsor/filter/mask.py:17-27,78-88`):
//     thresh   = 255 where alpha == 255                       (threshold(255 - a, 0, 255, BINARY_INV))
//     contours = findContours(thresh, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)
//     sorted by cx^2 + cy^2, (cx, cy) = int(m10/m00), int(m01/m00) of the contour *polygon*
//
// What the GPU needs per zone is the set of lattice points of the closed contour polygon: every
// contour edge runs in one of the 8 lattice directions and a detection box has integer corners, so
// "box polygon intersects zone polygon" (shapely, mask.py:45-54) <=> "a lattice point of the zone
// polygon lies in the closed box" (SURVEY.md a-7).  Those lattice points are the 8-connected
// alpha==255 component with its holes (and anything nested in them) filled -- RETR_EXTERNAL reports
// outermost borders only -- i.e. the 8-connected components of the complement of the background that
// is 4-connected to the image border.  The polygon itself is traced (Moore neighbour tracing of the
// outer border) only to reproduce the reference's ordering key with OpenCV's Green-formula moments.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/watsor_hip.h"

namespace {

struct Zone {
    int label;
    long long key;
    int cx, cy;
    int order;   // discovery order (raster scan of the first pixel)
};

// outer border of the region `lab == id`, starting at its raster-first pixel (sx, sy); 8-connectivity
void trace_border(const std::vector<int>& lab, int W, int H, int id, int sx, int sy,
                  std::vector<std::pair<int, int>>& pts) {
    static const int dx[8] = {1, 1, 0, -1, -1, -1, 0, 1};   // E, SE, S, SW, W, NW, N, NE (clockwise, y down)
    static const int dy[8] = {0, 1, 1, 1, 0, -1, -1, -1};
    auto in = [&](int x, int y) { return x >= 0 && y >= 0 && x < W && y < H && lab[(size_t)y * W + x] == id; };
    pts.clear();
    pts.emplace_back(sx, sy);
    // isolated pixel?
    bool any = false;
    for (int d = 0; d < 8; ++d) any |= in(sx + dx[d], sy + dy[d]);
    if (!any) return;
    // The start pixel is the top-most, left-most pixel: its W, NW, N, NE neighbours are background.
    int cx = sx, cy = sy;
    int back = 4;   // direction pointing at the background pixel we "came from" (west)
    int guard = 0;
    const int max_steps = 4 * (W * H + 4);
    while (guard++ < max_steps) {
        int found = -1;
        for (int k = 1; k <= 8; ++k) {   // scan clockwise starting just after the backtrack direction
            const int dd = (back + k) & 7;
            if (in(cx + dx[dd], cy + dy[dd])) { found = dd; break; }
        }
        if (found < 0) break;
        const int nx = cx + dx[found], ny = cy + dy[found];
        // closed: we are back on the start pixel and about to repeat the very first move
        if (cx == sx && cy == sy && pts.size() > 1 && nx == pts[1].first && ny == pts[1].second) {
            pts.pop_back();   // the closing visit of the start pixel duplicates pts[0]
            break;
        }
        // new backtrack = direction from the new pixel to the last background pixel examined
        const int prev = (found + 7) & 7;                 // last background neighbour (clockwise-before found)
        const int bx = cx + dx[prev], by = cy + dy[prev];
        int nb = 0;
        for (int dd = 0; dd < 8; ++dd)
            if (nx + dx[dd] == bx && ny + dy[dd] == by) { nb = dd; break; }
        back = nb;
        cx = nx;
        cy = ny;
        pts.emplace_back(cx, cy);
    }
}

}  // namespace

extern "C" int wz_zones_from_alpha(const uint8_t* alpha, int W, int H, int max_zones, uint8_t* zone_fill,
                                   int32_t* centroid_xy) {
    if (!alpha || !zone_fill || W < 1 || H < 1 || max_zones < 1) return WZ_EINVAL;
    const size_t N = (size_t)W * H;
    // 1. outside background: non-255 pixels 4-connected to the image border
    std::vector<uint8_t> outside(N, 0);
    std::vector<int> stack;
    auto push_if = [&](int x, int y) {
        const size_t i = (size_t)y * W + x;
        if (alpha[i] != 255 && !outside[i]) {
            outside[i] = 1;
            stack.push_back((int)i);
        }
    };
    for (int x = 0; x < W; ++x) { push_if(x, 0); push_if(x, H - 1); }
    for (int y = 0; y < H; ++y) { push_if(0, y); push_if(W - 1, y); }
    while (!stack.empty()) {
        const int i = stack.back();
        stack.pop_back();
        const int x = i % W, y = i / W;
        if (x > 0) push_if(x - 1, y);
        if (x < W - 1) push_if(x + 1, y);
        if (y > 0) push_if(x, y - 1);
        if (y < H - 1) push_if(x, y + 1);
    }
    // 2. filled zones: 8-connected components of !outside, labelled in raster order of first pixel
    std::vector<int> lab(N, 0);
    std::vector<Zone> zones;
    std::vector<std::pair<int, int>> pts;
    int nlab = 0;
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
            const size_t i = (size_t)y * W + x;
            if (outside[i] || lab[i]) continue;
            const int id = ++nlab;
            lab[i] = id;
            stack.push_back((int)i);
            while (!stack.empty()) {
                const int j = stack.back();
                stack.pop_back();
                const int px = j % W, py = j / W;
                for (int ddy = -1; ddy <= 1; ++ddy)
                    for (int ddx = -1; ddx <= 1; ++ddx) {
                        const int qx = px + ddx, qy = py + ddy;
                        if (qx < 0 || qy < 0 || qx >= W || qy >= H) continue;
                        const size_t q = (size_t)qy * W + qx;
                        if (!outside[q] && !lab[q]) {
                            lab[q] = id;
                            stack.push_back((int)q);
                        }
                    }
            }
            // 3. contour polygon moments (cv::moments on a contour: Green's formula, in double)
            trace_border(lab, W, H, id, x, y, pts);
            double a00 = 0, a10 = 0, a01 = 0;
            const size_t n = pts.size();
            double xi_1 = pts[n - 1].first, yi_1 = pts[n - 1].second;
            for (size_t k = 0; k < n; ++k) {
                const double xi = pts[k].first, yi = pts[k].second;
                const double dxy = xi_1 * yi - xi * yi_1;
                a00 += dxy;
                a10 += dxy * (xi_1 + xi);
                a01 += dxy * (yi_1 + yi);
                xi_1 = xi;
                yi_1 = yi;
            }
            double m00 = a00 * 0.5, m10 = a10 / 6.0, m01 = a01 / 6.0;
            if (a00 < 0) { m00 = -m00; m10 = -m10; m01 = -m01; }
            if (m00 == 0.0) return WZ_EFORMAT;   // the reference raises ZeroDivisionError here (mask.py:80)
            Zone z;
            z.label = id;
            z.cx = (int)(m10 / m00);             // Python int(): truncation toward zero
            z.cy = (int)(m01 / m00);
            z.key = (long long)z.cx * z.cx + (long long)z.cy * z.cy;
            z.order = id;
            zones.push_back(z);
        }
    }
    // 4. sorted(contours, key=contours_key): stable; OpenCV hands contours back last-found-first
    std::stable_sort(zones.begin(), zones.end(), [](const Zone& a, const Zone& b) {
        if (a.key != b.key) return a.key < b.key;
        return a.order > b.order;
    });
    if ((int)zones.size() > max_zones) return WZ_ELIMIT;
    std::vector<int> rank(nlab + 1, -1);
    for (size_t r = 0; r < zones.size(); ++r) {
        rank[zones[r].label] = (int)r;
        if (centroid_xy) {
            centroid_xy[2 * r] = zones[r].cx;
            centroid_xy[2 * r + 1] = zones[r].cy;
        }
    }
    memset(zone_fill, 0, N * zones.size());
    for (size_t i = 0; i < N; ++i)
        if (lab[i]) zone_fill[(size_t)rank[lab[i]] * N + i] = 1;
    return (int)zones.size();
}
