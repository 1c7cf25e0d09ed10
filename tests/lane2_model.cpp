// CPU model of bsw_lane2_kernel (TEST INFRASTRUCTURE): drives the product header bsw_lane2_core.h — the per-lane
// arithmetic the GPU kernel is compiled from — with the wave-level glue restated in plain loops (64 lanes x 2 seeds
// in lock step, wave-uniform minima/maxima, block dispatch).  tests/test_lane2_model.py checks it against the
// oracle, so the two-seeds-per-lane algebra is verified here, where no GPU exists.
// g++ -O2 -std=c++17 -shared -fPIC -I include -o lane2_model.so tests/lane2_model.cpp
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/bwa_sw_mi355.h"
#include "../bwa-mem-sw_amd/csrc/bsw_lane2_core.h"

using namespace bsw::l2;

// row storage of the looped kernel (lane2l): a plain array here, the accumulator register file on the GPU
template <int QMAX, int NW>
struct array_row {
    uint32_t eh[QMAX + 16];
    uint32_t rm[2][NW];
    void load8(int b, uint32_t (&T)[8]) const { for (int c = 0; c < 8; ++c) T[c] = eh[8 * b + c]; }
    void store8(int b, const uint32_t (&T)[8]) { for (int c = 0; c < 8; ++c) eh[8 * b + c] = T[c]; }
    uint32_t match_bytes(int b) const
    {
        const int wd = (b >> 2) < NW ? b >> 2 : NW - 1;             // (past the last block: any word, the value is unused)
        return bsw::l2::byte_pair_dyn(rm[0][wd], rm[1][wd], (uint32_t)b & 3u);
    }
    void swap8w(int b, uint32_t (&T)[8], uint32_t &Wc)
    {
        store8(b, T);
        for (int c = 0; c < 8; ++c) T[c] = eh[8 * b + 8 + c];
        Wc = match_bytes(b + 1);
    }
    void load8w(int b, uint32_t (&T)[8], uint32_t &Wc) const { load8(b, T); Wc = match_bytes(b); }
    void put_rm(int wd, uint32_t a, uint32_t b) { rm[0][wd] = a; rm[1][wd] = b; }
};

template <int QB, bool VM, bool SYM, bool LOOP>
struct wave_model {
    using LU = lane2<QB, VM, SYM>;
    using LL = lane2l<QB, VM, SYM>;
    using L = LU;
    struct lane_t {
        typename LU::state S;
        typename LL::state SL;
        array_row<LU::QMAX, LU::NW> row;
        uint32_t qp[2][3][L::NW];       // query bit planes
        uint32_t wn[L::NC];             // N planes, interleaved per 16 columns
        const uint8_t *t[2];
        bool valid[2];
        uint32_t ti[2];
    };

    static void run(const bsw_params *p, const bsw_task *tasks, int side, const uint32_t *order, size_t n, size_t w0,
                    const int32_t *h0s, bsw_ext *out)
    {
        consts k;
        k.a = p->mat[0]; k.pb = -p->mat[1]; k.pn = -p->mat[24];
        k.o_del = p->o_del; k.e_del = p->e_del; k.oe_ins = p->o_ins + p->e_ins; k.e_ins = p->e_ins; k.zdrop = p->zdrop;
        fill_packed_consts(k);
        int mx = 0;
        for (int i = 0; i < 25; ++i) mx = mx > p->mat[i] ? mx : p->mat[i];
        std::vector<lane_t> ln(64);
        uni u;
        u.nblk = 0;
        for (int l = 0; l < 64; ++l) {
            lane_t &a = ln[l];
            memset(a.qp, 0, sizeof(a.qp));
            for (int x = 0; x < 2; ++x) {
                const size_t slot = w0 + (size_t)l + 64 * (size_t)x;
                a.valid[x] = slot < n;
                a.ti[x] = a.valid[x] ? order[slot] : order[0];
                const bsw_task &T = tasks[a.ti[x]];
                int qlen = side ? T.rqlen : T.lqlen, tlen = side ? T.rtlen : T.ltlen;
                const uint8_t *q = side ? T.rquery : T.lquery;
                a.t[x] = side ? T.rtarget : T.ltarget;
                const int eb = side ? p->pen_clip3 : p->pen_clip5;
                int wl = side ? T.wlim_r : T.wlim_l;
                if (wl <= 0) {
                    int mi = (qlen * mx + eb - p->o_ins + p->e_ins) / p->e_ins, md = (qlen * mx + eb - p->o_del + p->e_del) / p->e_del;
                    if (mi < 1) mi = 1;
                    if (md < 1) md = 1;
                    wl = mi < md ? mi : md;
                }
                if (!a.valid[x]) tlen = 0;
                init_pair(a.S.p, x, qlen, tlen, h0s ? h0s[a.ti[x]] : T.h0, p->w < wl ? p->w : wl);
                a.SL.p = a.S.p;
                for (int j = 0; j < qlen; ++j) {
                    const int c = q[j] > 4 ? 4 : q[j];
                    if (c & 1) a.qp[x][0][j >> 5] |= 1u << (j & 31);
                    if (c & 2) a.qp[x][1][j >> 5] |= 1u << (j & 31);
                    if (c & 4) { a.qp[x][2][j >> 5] |= 1u << (j & 31); if (a.valid[x]) u.nblk |= 1u << (j >> 3); }
                }
            }
            for (int c = 0; c < L::NC; ++c) {
                const uint32_t wa = (a.qp[0][2][c >> 1] >> (16 * (c & 1))) & 0xffffu, wb = (a.qp[1][2][c >> 1] >> (16 * (c & 1))) & 0xffffu;
                a.wn[c] = wa | (wb << 16);
            }
            if constexpr (LOOP) { memset(&a.row, 0xa5, sizeof(a.row)); LL::init_row(a.SL, k, a.row); }
            else L::init_row(a.S, k);
        }
        std::vector<rowp> rv(64);
        for (int i = 0;; ++i) {
            bool any = false;
            u.jlo = 1 << 20; u.jhi = -1; u.jem = 1 << 20; u.anybite = false; u.zl = 1 << 20; u.zh = -1;
            for (int l = 0; l < 64; ++l) {
                pairv &pv = LOOP ? ln[l].SL.p : ln[l].S.p;
                row_begin2(pv, i, rv[l]);
                for (int x = 0; x < 2; ++x) {
                    if (!half_of(rv[l].ACT, x)) continue;
                    any = true;
                    const int beg = half_of(pv.BEG, x), end = half_of(pv.END, x);
                    if (beg < u.jlo) u.jlo = beg;
                    if (end > u.jhi) u.jhi = end;
                    if (end < u.jem) u.jem = end;
                    if (half_of(rv[l].BITE, x)) {
                        u.anybite = true;
                        if (half_of(rv[l].ZLO, x) < u.zl) u.zl = half_of(rv[l].ZLO, x);
                        if (half_of(rv[l].ZHI, x) > u.zh) u.zh = half_of(rv[l].ZHI, x);
                    }
                }
            }
            if (!any) break;
            for (int l = 0; l < 64; ++l) {
                lane_t &a = ln[l];
                int tb[2];
                for (int x = 0; x < 2; ++x) {
                    int b = half_of(rv[l].ACT, x) ? a.t[x][i] : 0;
                    tb[x] = b > 4 ? 4 : b;
                }
                auto qp = [&](int x, int b, uint32_t (&rm)[L::NW]) {
                    for (int wd = 0; wd < L::NW; ++wd) rm[wd] = L::base_match(a.qp[x][0][wd], a.qp[x][1][wd], a.qp[x][2][wd], b);
                };
                auto wn = [&](int c) { return a.wn[c]; };
                auto kp = [&](int b, uint32_t (&kw)[L::NW]) {
                    for (int wd = 0; wd < L::NW; ++wd) kw[wd] = L::keep_word(b, wd);
                };
                if constexpr (LOOP) LL::row_body(a.SL, k, i, rv[l], u, tb, qp, kp, wn, a.row);
                else L::row_body(a.S, k, i, rv[l], u, tb, qp, kp, wn);
            }
        }
        for (int l = 0; l < 64; ++l)
            for (int x = 0; x < 2; ++x) {
                if (!ln[l].valid[x]) continue;
                const ext_out s = pair_result(LOOP ? ln[l].SL.p : ln[l].S.p, x);
                bsw_ext &e = out[ln[l].ti[x]];
                e.score = s.mx; e.qle = s.max_j + 1; e.tle = s.max_i + 1; e.gtle = s.max_ie + 1;
                e.gscore = s.gscore; e.max_off = s.max_off; e.aw = p->w; e.cells = s.cells;
            }
    }
};

// One band try of one side of tasks[order[0..n)] exactly as bsw_lane2_kernel<QB> would run it: 128 seeds per wave; qb = 17 (the
// 136-column class) or 9 (the 72-column class that runs at three waves per SIMD).
// h0s: optional per-task h0 override (the score after the left extension); out is indexed by task.
extern "C" int lane2_model_run_qb(const bsw_params *p, const bsw_task *tasks, int side, const uint32_t *order, size_t n,
                                  const int32_t *h0s, bsw_ext *out, int qb)
{
    if (!p || !tasks || !order || !out) return -1;
    if (p->mat[1] > 0 || p->mat[24] > 0 || -p->mat[1] < -p->mat[24]) return -2;      /* lane2_params_ok */
    if (p->o_del + p->e_del > 255 || p->o_ins + p->e_ins > 255 || p->mat[0] - p->mat[1] > 255) return -2;
    const bool sym = p->o_del == p->o_ins && p->e_del == p->e_ins, vm = p->variant == BSW_VARIANT_M;
    for (size_t w0 = 0; w0 < n; w0 += 128) {
#define RUNU(QB) do { \
        if (!vm && sym) wave_model<QB, false, true, false>::run(p, tasks, side, order, n, w0, h0s, out); \
        else if (!vm) wave_model<QB, false, false, false>::run(p, tasks, side, order, n, w0, h0s, out); \
        else if (sym) wave_model<QB, true, true, false>::run(p, tasks, side, order, n, w0, h0s, out); \
        else wave_model<QB, true, false, false>::run(p, tasks, side, order, n, w0, h0s, out); } while (0)
        if (qb == 17) RUNU(17); else if (qb == 9) RUNU(9); else return -3;
#undef RUNU
    }
    return 0;
}
extern "C" int lane2_model_run(const bsw_params *p, const bsw_task *tasks, int side, const uint32_t *order, size_t n,
                               const int32_t *h0s, bsw_ext *out)
{
    return lane2_model_run_qb(p, tasks, side, order, n, h0s, out, 17);
}

// The looped kernel (bsw_lane2l_kernel<QB>: blocks walked by a run-time loop, row behind an accessor), qb = 17 or 29.
extern "C" int lane2l_model_run(const bsw_params *p, const bsw_task *tasks, int side, const uint32_t *order, size_t n,
                                const int32_t *h0s, bsw_ext *out, int qb)
{
    if (!p || !tasks || !order || !out) return -1;
    if (p->mat[1] > 0 || p->mat[24] > 0 || -p->mat[1] < -p->mat[24]) return -2;
    if (p->o_del + p->e_del > 255 || p->o_ins + p->e_ins > 255 || p->mat[0] - p->mat[1] > 255) return -2;
    const bool sym = p->o_del == p->o_ins && p->e_del == p->e_ins, vm = p->variant == BSW_VARIANT_M;
    for (size_t w0 = 0; w0 < n; w0 += 128) {
#define RUNL(QB) do { \
        if (!vm && sym) wave_model<QB, false, true, true>::run(p, tasks, side, order, n, w0, h0s, out); \
        else if (!vm) wave_model<QB, false, false, true>::run(p, tasks, side, order, n, w0, h0s, out); \
        else if (sym) wave_model<QB, true, true, true>::run(p, tasks, side, order, n, w0, h0s, out); \
        else wave_model<QB, true, false, true>::run(p, tasks, side, order, n, w0, h0s, out); } while (0)
        if (qb == 17) RUNL(17); else if (qb == 29) RUNL(29); else return -3;
#undef RUNL
    }
    return 0;
}
