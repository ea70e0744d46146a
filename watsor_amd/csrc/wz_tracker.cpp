// Recurring-detection tracker and the sieve's row rewrite, host C++.
//
// Replaces, for one camera, what the reference's sieve thread does per frame in Python:
//   DetectionSieve._incoming_frame   watsor/filter/sieve.py:21-33   (clone 100 rows, run the filters, write back, zero the rest)
//   TrackFilter.__call__ / _group_and_update / _centroid / _combine   watsor/filter/track.py:25-149
// The per-detection filters (`label > 0 and Confidence and Area and Mask`, track.py:26) have already been evaluated on
// the GPU by wz_k_rows: either the caller hands over the pass bytes, or the camera is in drop mode (wz_set_camera_drop)
// and failing rows arrive with label 0.
//
// Everything here is at most 100 x 100 per frame, stateful and sequential: it is host code on purpose.
//
// Two behaviours of the reference come from its runtime, not from its source (oracle/tracker.py has the details):
//   * new tracks are appended, and a combined row's zones are listed, in the iteration order of a CPython `set` of
//     small ints.  `PySmallIntSet` below reproduces CPython's open-addressing table (hash(i) = i, 9 linear probes,
//     perturb shift 5, growth to 4x used at 3/5 load -- Objects/setobject.c, unchanged from 3.7 to 3.12) so that the
//     order is the one a CPython reference produces; tests compare it with real `set`s.
//   * `np.argsort` (track.py:67) is unstable: which of several EQUALLY near tracks is visited first is defined by the
//     numpy build.  Here: index order (a stable sort).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <deque>
#include <new>
#include <vector>

#include "wz_common.h"

namespace {

// CPython set of non-negative ints: add (no removal), growth, iteration in slot order.
class PySmallIntSet {
  public:
    PySmallIntSet() : table_(8, kEmpty), mask_(7), fill_(0) {}

    bool add(int64_t key) {                                   // set_add_entry
        size_t perturb = (size_t)key;
        size_t i = (size_t)key & mask_;
        size_t e;
        for (;;) {
            e = i;
            int probes = (i + kLinearProbes <= mask_) ? kLinearProbes : 0;
            do {
                if (table_[e] == kEmpty) goto found_unused;
                if (table_[e] == key) return false;
                ++e;
            } while (probes--);
            perturb >>= kPerturbShift;
            i = (i * 5 + 1 + perturb) & mask_;
        }
    found_unused:
        table_[e] = key;
        ++fill_;
        if (fill_ * 5 < mask_ * 3) return true;
        resize(fill_ > 50000 ? fill_ * 2 : fill_ * 4);
        return true;
    }

    template <typename F>
    void for_each(F f) const {
        for (size_t s = 0; s <= mask_; ++s)
            if (table_[s] != kEmpty) f(table_[s]);
    }
    size_t size() const { return fill_; }

  private:
    static constexpr int64_t kEmpty = -1;
    static constexpr int kLinearProbes = 9;
    static constexpr int kPerturbShift = 5;

    void resize(size_t minused) {                             // set_table_resize + set_insert_clean
        size_t newsize = 8;
        while (newsize <= minused) newsize <<= 1;
        std::vector<int64_t> old;
        old.swap(table_);
        table_.assign(newsize, kEmpty);
        mask_ = newsize - 1;
        for (int64_t key : old) {
            if (key == kEmpty) continue;
            size_t perturb = (size_t)key;
            size_t i = (size_t)key & mask_;
            for (;;) {
                size_t e = i;
                if (table_[e] == kEmpty) { table_[e] = key; break; }
                bool placed = false;
                if (i + kLinearProbes <= mask_) {
                    for (int j = 0; j < kLinearProbes; ++j) {
                        ++e;
                        if (table_[e] == kEmpty) { table_[e] = key; placed = true; break; }
                    }
                }
                if (placed) break;
                perturb >>= kPerturbShift;
                i = (i * 5 + 1 + perturb) & mask_;
            }
        }
    }

    std::vector<int64_t> table_;
    size_t mask_, fill_;
};

// `set(range(n)).difference(used)` iterated (track.py:89,97): CPython copies and discards (ascending order) when
// len(so) / 4 > len(other), otherwise inserts the survivors one by one into a new set (slot order of that set).
void unused_in_set_order(int n, const std::vector<char>& used, int n_used, std::vector<int>& out) {
    out.clear();
    if ((n >> 2) > n_used) {
        for (int i = 0; i < n; ++i)
            if (!used[i]) out.push_back(i);
        return;
    }
    PySmallIntSet s;
    for (int i = 0; i < n; ++i)
        if (!used[i]) s.add(i);
    s.for_each([&](int64_t k) { out.push_back((int)k); });
}

typedef std::deque<wz_detection_t> History;                   // deque(maxlen=history): oldest first

struct LabelTracks {
    int32_t label;
    std::vector<History> tracks;
};

inline void centroid(const wz_detection_t& d, int64_t& cx, int64_t& cy) {      // track.py:112-116
    cx = (int64_t)(((double)((int64_t)d.x_min + (int64_t)d.x_max)) / 2.0);
    cy = (int64_t)(((double)((int64_t)d.y_min + (int64_t)d.y_max)) / 2.0);
}

void combine(const History& h, wz_detection_t& o) {            // track.py:118-149
    memset(&o, 0, sizeof(o));
    const wz_detection_t& f = h[0];
    o.label = f.label;
    o.confidence = f.confidence;
    o.x_min = f.x_min;
    o.y_min = f.y_min;
    o.x_max = f.x_max;
    o.y_max = f.y_max;
    for (size_t i = 1; i < h.size(); ++i) {
        const wz_detection_t& d = h[i];
        if (d.confidence > o.confidence) o.confidence = d.confidence;          // max(a, b): b only if b > a
        if (d.x_min < o.x_min) o.x_min = d.x_min;
        if (d.y_min < o.y_min) o.y_min = d.y_min;
        if (d.x_max > o.x_max) o.x_max = d.x_max;
        if (d.y_max > o.y_max) o.y_max = d.y_max;
    }
    PySmallIntSet zones;
    for (const wz_detection_t& d : h)
        for (int z = 0; z < WZ_MAX_ZONES; ++z)
            if (d.zones[z] > 0) zones.add(d.zones[z]);
    int k = 0;
    zones.for_each([&](int64_t z) {
        if (k < WZ_MAX_ZONES) o.zones[k++] = (int32_t)z;
    });
}

}  // namespace

struct wz_tracker {
    int sensitivity, history;
    std::vector<LabelTracks> by_label;                        // insertion-ordered, like the reference's dict
    // scratch, reused across frames
    std::vector<std::pair<int32_t, std::vector<int>>> groups;
    std::vector<double> dist, rmin;
    std::vector<int> order, nearest, fresh;
    std::vector<char> used_k, used_i;
    std::vector<int64_t> cin, ckn;
    std::vector<wz_detection_t> result;
};

static void tracker_step(wz_tracker* t, const wz_detection_t* rows, int n, const uint8_t* pass, int* suspicious) {
    // group by label in order of first appearance (track.py:31-33)
    size_t ng = 0;
    for (int i = 0; i < n; ++i) {
        if (!(rows[i].label > 0) || (pass && !pass[i])) continue;
        size_t g = 0;
        while (g < ng && t->groups[g].first != rows[i].label) ++g;
        if (g == ng) {
            if (t->groups.size() <= ng) t->groups.emplace_back();
            t->groups[ng].first = rows[i].label;
            t->groups[ng].second.clear();
            ++ng;
        }
        t->groups[g].second.push_back(i);
    }
    *suspicious = ng > 0 ? 1 : 0;                             // track.py:38

    // labels that are no longer detected (track.py:41-46)
    t->by_label.erase(std::remove_if(t->by_label.begin(), t->by_label.end(),
                                     [&](const LabelTracks& lt) {
                                         for (size_t g = 0; g < ng; ++g)
                                             if (t->groups[g].first == lt.label) return false;
                                         return true;
                                     }),
                      t->by_label.end());

    for (size_t g = 0; g < ng; ++g) {
        const int32_t label = t->groups[g].first;
        const std::vector<int>& in = t->groups[g].second;
        size_t li = 0;
        while (li < t->by_label.size() && t->by_label[li].label != label) ++li;
        if (li == t->by_label.size()) {                       // defaultdict access inserts (track.py:56)
            t->by_label.emplace_back();
            t->by_label.back().label = label;
        }
        std::vector<History>& known = t->by_label[li].tracks;
        const int n_in = (int)in.size(), n_known = (int)known.size();

        t->cin.resize(2 * (size_t)n_in);
        t->ckn.resize(2 * (size_t)n_known);
        for (int i = 0; i < n_in; ++i) centroid(rows[in[i]], t->cin[2 * i], t->cin[2 * i + 1]);
        for (int k = 0; k < n_known; ++k) centroid(known[k][0], t->ckn[2 * k], t->ckn[2 * k + 1]);

        t->order.clear();
        t->nearest.assign(n_known, 0);
        if (n_known > 0 && n_in > 0) {
            // euclidean cdist in double (track.py:63), row minima, first arg-minimum (track.py:67-70)
            t->rmin.resize(n_known);
            for (int k = 0; k < n_known; ++k) {
                double best = 0.0;
                int arg = 0;
                for (int i = 0; i < n_in; ++i) {
                    const double dx = (double)t->ckn[2 * k] - (double)t->cin[2 * i];
                    const double dy = (double)t->ckn[2 * k + 1] - (double)t->cin[2 * i + 1];
                    const double sq = dx * dx;
                    const double d = std::sqrt(sq + dy * dy);
                    if (i == 0 || d < best) {
                        best = d;
                        arg = i;
                    }
                }
                t->rmin[k] = best;
                t->nearest[k] = arg;
            }
            t->order.resize(n_known);
            for (int k = 0; k < n_known; ++k) t->order[k] = k;
            std::stable_sort(t->order.begin(), t->order.end(),
                             [&](int a, int b) { return t->rmin[a] < t->rmin[b]; });
        }

        // each track claims only its nearest input; first come first served (track.py:75-86)
        t->used_k.assign(n_known, 0);
        t->used_i.assign(n_in, 0);
        int n_used = 0;
        for (int k : t->order) {
            const int i = t->nearest[k];
            if (t->used_k[k] || t->used_i[i]) continue;
            known[k].push_back(rows[in[i]]);
            if ((int)known[k].size() > t->history) known[k].pop_front();
            t->used_k[k] = 1;
            t->used_i[i] = 1;
            ++n_used;
        }
        for (int k = n_known - 1; k >= 0; --k)                // track.py:92-94
            if (!t->used_k[k]) known.erase(known.begin() + k);
        unused_in_set_order(n_in, t->used_i, n_used, t->fresh);
        for (int i : t->fresh) {                              // track.py:97-99
            known.emplace_back();
            known.back().push_back(rows[in[i]]);
        }
    }

    t->result.clear();
    for (const LabelTracks& lt : t->by_label)                 // track.py:103-110
        for (const History& h : lt.tracks) {
            if ((int)h.size() < t->sensitivity) continue;
            t->result.emplace_back();
            combine(h, t->result.back());
        }
}

extern "C" {

int wz_tracker_create(int sensitivity, int history, wz_tracker_t** out) {
    if (!out) return wz_set_error(WZ_EINVAL, "wz_tracker_create: out is NULL");
    *out = nullptr;
    if (history < 1)
        return wz_set_error(WZ_EINVAL, "wz_tracker_create: history %d < 1 (a deque(maxlen=0) track cannot be matched)",
                            history);
    wz_tracker* t = new (std::nothrow) wz_tracker();
    if (!t) return wz_set_error(WZ_EINVAL, "wz_tracker_create: out of memory");
    t->sensitivity = sensitivity;
    t->history = history;
    *out = t;
    return 0;
}

void wz_tracker_destroy(wz_tracker_t* t) { delete t; }

int wz_tracker_reset(wz_tracker_t* t) {
    if (!t) return wz_set_error(WZ_EINVAL, "wz_tracker_reset: NULL tracker");
    t->by_label.clear();
    return 0;
}

int wz_tracker_count(wz_tracker_t* t) {
    if (!t) return wz_set_error(WZ_EINVAL, "wz_tracker_count: NULL tracker");
    int c = 0;
    for (const LabelTracks& lt : t->by_label) c += (int)lt.tracks.size();
    return c;
}

int wz_tracker_update(wz_tracker_t* t, const wz_detection_t* rows, int n, const uint8_t* pass, wz_detection_t* out,
                      int cap, int* n_out, int* suspicious) {
    if (!t || n < 0 || (n > 0 && !rows) || cap < 0 || (cap > 0 && !out))
        return wz_set_error(WZ_EINVAL, "wz_tracker_update: bad argument");
    int sa = 0;
    tracker_step(t, rows, n, pass, &sa);
    const int total = (int)t->result.size();
    const int k = std::min(total, cap);
    if (k > 0) memcpy(out, t->result.data(), sizeof(wz_detection_t) * (size_t)k);
    if (n_out) *n_out = total;
    if (suspicious) *suspicious = sa;
    return 0;
}

int wz_tracker_sieve(wz_tracker_t* t, wz_detection_t* rows, int n, const uint8_t* pass, int* suspicious) {
    if (!t || n < 0 || (n > 0 && !rows)) return wz_set_error(WZ_EINVAL, "wz_tracker_sieve: bad argument");
    int sa = 0;
    tracker_step(t, rows, n, pass, &sa);                      // histories hold copies: rows may be overwritten now
    const int k = std::min((int)t->result.size(), n);         // sieve.py:47-56: results first, the rest zeroed
    if (k > 0) memcpy(rows, t->result.data(), sizeof(wz_detection_t) * (size_t)k);
    if (n > k) memset(rows + k, 0, sizeof(wz_detection_t) * (size_t)(n - k));
    if (suspicious) *suspicious = sa;
    return 0;
}

#ifdef WZ_DEV_BUILD   // test hooks of the CPython-set emulation: development library only
int wz_debug_pyset_order(const int32_t* keys, int n, int32_t* out) {
    if (n < 0 || (n > 0 && (!keys || !out))) return wz_set_error(WZ_EINVAL, "wz_debug_pyset_order: bad argument");
    PySmallIntSet s;
    for (int i = 0; i < n; ++i) {
        if (keys[i] < 0) return wz_set_error(WZ_EINVAL, "wz_debug_pyset_order: negative key");
        s.add(keys[i]);
    }
    int k = 0;
    s.for_each([&](int64_t v) { out[k++] = (int32_t)v; });
    return k;
}

int wz_debug_unused_order(int n, const uint8_t* used, int32_t* out) {
    if (n < 0 || (n > 0 && (!used || !out))) return wz_set_error(WZ_EINVAL, "wz_debug_unused_order: bad argument");
    std::vector<char> u(used, used + n);
    int n_used = 0;
    for (int i = 0; i < n; ++i) n_used += u[i] ? 1 : 0;
    std::vector<int> o;
    unused_in_set_order(n, u, n_used, o);
    for (size_t i = 0; i < o.size(); ++i) out[i] = o[i];
    return (int)o.size();
}
#endif

}  // extern "C"
