// Host-side (no GPU) integer table construction and error plumbing of libstep_amd.so.
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <tuple>
#include <vector>

#include "../../include/step_amd.h"

static thread_local char g_err[512] = "";

void step_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* step_last_error(void) { return g_err; }
extern "C" int step_abi_version(void) { return 12; }   // 12: mri_*_any; 11: mri_step_update; 10: mri_loss_grad_mc_fused; 9: bh_update_t.eps; 8: mri_loss_grad_split; 7: nufft_plan_t lane-per-chain spread lists; 6: nufft_plan_t owner bins, nufft_fwd_grid / nufft_inv_grid; 5: bh_update_t.mcmc, mcmc_update_philox quad_w; 4: bh_loss_grad_sep3 takes `update`

#define REQ(cond, ...)                      \
    do {                                    \
        if (!(cond)) {                      \
            step_set_error(__VA_ARGS__);    \
            return STEP_E_INVALID;          \
        }                                   \
    } while (0)

// ------------------------------------------------------------------ closure tables
// Restates the dictionary logic of forward_operator/blackhole.py:235-264: key = (time, a, b)
// with time compared as an exact float64 (python dict hashing of the numpy value), later rows
// overwrite earlier ones, the swapped key maps to the same row with conj = 1.
namespace {
struct Key {
    uint64_t tbits;
    int32_t a, b;
    bool operator<(const Key& o) const {
        return std::tie(tbits, a, b) < std::tie(o.tbits, o.a, o.b);
    }
};
inline uint64_t time_bits(double t) {
    if (t == 0.0) t = 0.0;   // -0.0 == 0.0 as dict keys
    uint64_t u;
    memcpy(&u, &t, 8);
    return u;
}
}  // namespace

extern "C" int bh_build_closure_tables(const double* row_time, const int32_t* row_t1,
                                       const int32_t* row_t2, int64_t n_rows,
                                       const double* tri_time, const int32_t* tri_sites,
                                       int64_t n_tri, const double* quad_time,
                                       const int32_t* quad_sites, int64_t n_quad,
                                       int64_t* cp_index, int64_t* cp_conj, int64_t* camp_index,
                                       int64_t* camp_conj) {
    REQ(row_time && row_t1 && row_t2 && n_rows > 0, "bh_build_closure_tables: empty row table");
    REQ(n_tri >= 0 && n_quad >= 0, "bh_build_closure_tables: negative count");
    REQ(n_tri == 0 || (tri_time && tri_sites && cp_index && cp_conj), "bh_build_closure_tables: null triangle arrays");
    REQ(n_quad == 0 || (quad_time && quad_sites && camp_index && camp_conj), "bh_build_closure_tables: null quadrangle arrays");
    std::map<Key, std::pair<int64_t, int64_t>> rows;   // -> (row, conj)
    for (int64_t i = 0; i < n_rows; ++i) {
        const uint64_t tb = time_bits(row_time[i]);
        rows[Key{tb, row_t1[i], row_t2[i]}] = {i, 0};
        rows[Key{tb, row_t2[i], row_t1[i]}] = {i, 1};
    }
    auto look = [&](uint64_t tb, int32_t a, int32_t b, int64_t* idx, int64_t* cj) -> bool {
        auto it = rows.find(Key{tb, a, b});
        if (it == rows.end()) return false;
        *idx = it->second.first;
        *cj = it->second.second;
        return true;
    };
    for (int64_t k = 0; k < n_tri; ++k) {
        const uint64_t tb = time_bits(tri_time[k]);
        const int32_t* s = tri_sites + 3 * k;
        const int32_t leg[3][2] = {{s[0], s[1]}, {s[1], s[2]}, {s[2], s[0]}};
        for (int j = 0; j < 3; ++j)
            REQ(look(tb, leg[j][0], leg[j][1], &cp_index[3 * k + j], &cp_conj[3 * k + j]),
                "bh_build_closure_tables: triangle %lld leg %d: no baseline (%d,%d) at time %.17g",
                (long long)k, j, leg[j][0], leg[j][1], tri_time[k]);
    }
    for (int64_t k = 0; k < n_quad; ++k) {
        const uint64_t tb = time_bits(quad_time[k]);
        const int32_t* s = quad_sites + 4 * k;
        const int32_t leg[4][2] = {{s[0], s[1]}, {s[2], s[3]}, {s[0], s[3]}, {s[1], s[2]}};
        for (int j = 0; j < 4; ++j)
            REQ(look(tb, leg[j][0], leg[j][1], &camp_index[4 * k + j], &camp_conj[4 * k + j]),
                "bh_build_closure_tables: quadrangle %lld leg %d: no baseline (%d,%d) at time %.17g",
                (long long)k, j, leg[j][0], leg[j][1], quad_time[k]);
    }
    return STEP_OK;
}

extern "C" int bh_build_incidence(const int64_t* cp_index, const int64_t* cp_conj, int64_t n_tri,
                                  const int64_t* camp_index, const int64_t* camp_conj,
                                  int64_t n_quad, int64_t n_vis, int32_t* cp_ptr, int32_t* cp_ent,
                                  int32_t* ca_ptr, int32_t* ca_ent) {
    REQ(n_vis > 0 && n_tri >= 0 && n_quad >= 0, "bh_build_incidence: bad sizes");
    REQ(cp_ptr && ca_ptr, "bh_build_incidence: null ptr arrays");
    REQ(n_tri < (1 << 28) && n_quad < (1 << 28), "bh_build_incidence: too many closures");
    auto build = [&](const int64_t* index, const int64_t* conj, int64_t n, int legs, int32_t* ptr,
                     int32_t* ent, const char* what) -> int {
        std::vector<int32_t> count(n_vis + 1, 0);
        for (int64_t e = 0; e < n * legs; ++e) {
            REQ(index[e] >= 0 && index[e] < n_vis, "bh_build_incidence: %s index %lld out of range [0,%lld)",
                what, (long long)index[e], (long long)n_vis);
            REQ(conj[e] == 0 || conj[e] == 1, "bh_build_incidence: %s conjugate flag %lld not 0/1", what, (long long)conj[e]);
            count[index[e] + 1]++;
        }
        ptr[0] = 0;
        for (int64_t m = 0; m < n_vis; ++m) ptr[m + 1] = ptr[m] + count[m + 1];
        std::vector<int32_t> fill(ptr, ptr + n_vis);
        for (int64_t k = 0; k < n; ++k)          // ascending k per baseline: fixed summation order
            for (int j = 0; j < legs; ++j) {
                const int64_t e = k * legs + j;
                ent[fill[index[e]]++] = (int32_t)((k << 3) | (j << 1) | conj[e]);
            }
        return STEP_OK;
    };
    if (n_tri > 0) {
        REQ(cp_index && cp_conj && cp_ent, "bh_build_incidence: null triangle arrays");
        if (int rc = build(cp_index, cp_conj, n_tri, 3, cp_ptr, cp_ent, "cp")) return rc;
    } else {
        for (int64_t m = 0; m <= n_vis; ++m) cp_ptr[m] = 0;
    }
    if (n_quad > 0) {
        REQ(camp_index && camp_conj && ca_ent, "bh_build_incidence: null quadrangle arrays");
        if (int rc = build(camp_index, camp_conj, n_quad, 4, ca_ptr, ca_ent, "camp")) return rc;
    } else {
        for (int64_t m = 0; m <= n_vis; ++m) ca_ptr[m] = 0;
    }
    return STEP_OK;
}

// Per-frame closure lists for the fused step (bh_loss_grad_sep3): slot j of frame t holds one
// triangle / quadrangle whose baselines all lie in [t F, (t + 1) F), with local indices.
extern "C" int bh_build_frame_closures(const int64_t* cp_index, const int64_t* cp_conj, int64_t n_tri,
                                       const int64_t* camp_index, int64_t n_quad, int T, int F,
                                       int32_t* frame_tab) {
    REQ(T > 0 && F > 0 && F <= 31 && n_tri >= 0 && n_quad >= 0 && frame_tab, "bh_build_frame_closures: bad arguments");
    REQ(n_tri == 0 || (cp_index && cp_conj), "bh_build_frame_closures: null triangle tables");
    REQ(n_quad == 0 || camp_index, "bh_build_frame_closures: null quadrangle tables");
    for (int64_t i = 0; i < (int64_t)T * 2 * 64; ++i) { frame_tab[2 * i] = -1; frame_tab[2 * i + 1] = 0; }
    std::vector<int> fill(2 * (size_t)T, 0);
    auto place = [&](const int64_t* idx, const int64_t* conj, int legs, int64_t k, int kind) -> int {
        const int64_t t = idx[0] / F;
        REQ(idx[0] >= 0 && t < T, "bh_build_frame_closures: baseline index %lld out of range", (long long)idx[0]);
        int32_t packed = 0;
        for (int j = 0; j < legs; ++j) {
            REQ(idx[j] >= t * F && idx[j] < (t + 1) * F,
                "bh_build_frame_closures: %s %lld spans frames (baselines %lld and %lld)",
                kind ? "quadrangle" : "triangle", (long long)k, (long long)idx[0], (long long)idx[j]);
            packed |= (int32_t)(idx[j] - t * F) << (5 * j);
            if (conj) {
                REQ(conj[j] == 0 || conj[j] == 1, "bh_build_frame_closures: conjugate flag %lld", (long long)conj[j]);
                packed |= (int32_t)conj[j] << (15 + j);
            }
        }
        int& n = fill[2 * t + kind];
        REQ(n < 64, "bh_build_frame_closures: frame %lld has more than 64 %s", (long long)t,
            kind ? "quadrangles" : "triangles");
        int32_t* e = frame_tab + ((t * 2 + kind) * 64 + n) * 2;
        e[0] = (int32_t)k;
        e[1] = packed;
        ++n;
        return STEP_OK;
    };
    for (int64_t k = 0; k < n_tri; ++k)
        if (int rc = place(cp_index + 3 * k, cp_conj + 3 * k, 3, k, 0)) return rc;
    for (int64_t k = 0; k < n_quad; ++k)
        if (int rc = place(camp_index + 4 * k, nullptr, 4, k, 1)) return rc;
    return STEP_OK;
}

// ------------------------------------------------------------------ MRI mask
// numpy's legacy global generator, as used by dynamic_mri.py:28,33:
// np.random.seed(42) = MT19937 init_genrand(42); np.random.shuffle(a) on a 1-D array =
// for i in n-1..1: j = random_interval(i); swap(a[i], a[j]), where random_interval draws
// 32-bit words masked to the smallest 2^k-1 >= i and rejects values > i.
namespace {
struct MT19937 {
    uint32_t mt[624];
    int pos;
    explicit MT19937(uint32_t seed) {
        mt[0] = seed;
        for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        pos = 624;
    }
    void gen() {
        for (int k = 0; k < 624; ++k) {
            const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
            mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        pos = 0;
    }
    uint32_t next() {
        if (pos == 624) gen();
        uint32_t y = mt[pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    uint32_t interval(uint32_t max) {
        if (max == 0) return 0;
        uint32_t mask = max;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        uint32_t v;
        while ((v = next() & mask) > max) {}
        return v;
    }
};
}  // namespace

extern "C" int mri_build_mask(int acc_factor, int mode, int resolution, int frames,
                              int num_acs_lines, float* mask) {
    REQ(mask, "mri_build_mask: null mask");
    REQ(acc_factor > 0 && resolution > 0 && frames > 0 && num_acs_lines >= 0,
        "mri_build_mask: bad arguments acc=%d res=%d frames=%d acs=%d", acc_factor, resolution, frames, num_acs_lines);
    REQ(mode >= 0 && mode <= 2, "Unknown mode: %d", mode);
    const int half = resolution / 2;
    // python slice [half - acs//2 : half + acs//2] with negative-start semantics
    auto norm = [&](int v) { if (v < 0) v += resolution; return std::min(std::max(v, 0), resolution); };
    const int lo = norm(half - num_acs_lines / 2), hi = norm(half + num_acs_lines / 2);
    std::vector<int> keep(resolution, 0);
    for (int r = 0; r < resolution; r += acc_factor) keep[r] = 1;
    for (int r = lo; r < hi; ++r) keep[r] = 1;
    if (mode == 0) {
        for (int r = 0; r < resolution; ++r) mask[r] = (float)keep[r];
        return STEP_OK;
    }
    std::vector<int> lines;
    for (int r = 0; r < resolution; ++r) if (keep[r]) lines.push_back(r);
    if (mode == 2) {
        MT19937 rng(42u);
        for (int i = (int)lines.size() - 1; i >= 1; --i) {
            const int j = (int)rng.interval((uint32_t)i);
            std::swap(lines[i], lines[j]);
        }
    }
    // np.array_split: the first (n % frames) parts get n/frames + 1 entries
    for (size_t e = 0; e < (size_t)frames * resolution; ++e) mask[e] = 0.f;
    const int n = (int)lines.size(), q = n / frames, rem = n % frames;
    int pos = 0;
    for (int f = 0; f < frames; ++f) {
        const int cnt = q + (f < rem ? 1 : 0);
        for (int c = 0; c < cnt; ++c) mask[(size_t)f * resolution + lines[pos++]] = 1.f;
    }
    return STEP_OK;
}
