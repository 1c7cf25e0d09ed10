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

// ---- bsw_lane2g_kernel: a seed pair per GROUP of eight lanes (lane2g in bsw_lane2_core.h).  The per-lane phases are the
// product header's; the exchanges between them (the max-plus scan of the f offers, the H(i, j0 - 1) shift, the group
// reductions) are restated on arrays.  One wavefront = 8 groups = 16 seeds: slot 16 w + k (low halves) and 16 w + 8 + k.
template <int NS, bool VM, bool SYM>
struct group_model {
    using L = lane2g<NS, VM, SYM>;
    struct lane_t {
        typename L::state S;
        uint32_t mA[NS], mB[NS], WNs[NS];
    };
    static uint32_t sat8(int v) { return dup16((v > 255 ? 255 : v) << 8); }

    static void run(const bsw_params *p, const bsw_task *tasks, int side, const uint32_t *order, size_t n, size_t w0,
                    const int32_t *h0s, bsw_ext *out)
    {
        consts k;
        k.a = p->mat[0]; k.pb = -p->mat[1]; k.pn = -p->mat[24];
        k.o_del = p->o_del; k.e_del = p->e_del; k.oe_ins = p->o_ins + p->e_ins; k.e_ins = p->e_ins; k.zdrop = p->zdrop;
        fill_packed_consts(k);
        int mx = 0;
        for (int i = 0; i < 25; ++i) mx = mx > p->mat[i] ? mx : p->mat[i];
        const uint32_t E8 = sat8(8 * k.e_ins), E16 = sat8(16 * k.e_ins), E32 = sat8(32 * k.e_ins);
        std::vector<lane_t> ln(64);
        const uint8_t *tq[8][2];
        bool valid[8][2];
        uint32_t ti[8][2];
        bool nqs[NS] = {false};
        for (int grp = 0; grp < 8; ++grp)
            for (int x = 0; x < 2; ++x) {
                const size_t slot = w0 + (size_t)grp + 8 * (size_t)x;
                valid[grp][x] = slot < n;
                ti[grp][x] = valid[grp][x] ? order[slot] : order[0];
                const bsw_task &T = tasks[ti[grp][x]];
                int qlen = side ? T.rqlen : T.lqlen, tlen = side ? T.rtlen : T.ltlen;
                const uint8_t *q = side ? T.rquery : T.lquery;
                tq[grp][x] = side ? T.rtarget : T.ltarget;
                const int eb = side ? p->pen_clip3 : p->pen_clip5;
                int wl = side ? T.wlim_r : T.wlim_l;
                if (wl <= 0) {
                    int mi = (qlen * mx + eb - p->o_ins + p->e_ins) / p->e_ins, md = (qlen * mx + eb - p->o_del + p->e_del) / p->e_del;
                    if (mi < 1) mi = 1;
                    if (md < 1) md = 1;
                    wl = mi < md ? mi : md;
                }
                if (!valid[grp][x]) tlen = 0;
                for (int g = 0; g < 8; ++g) {
                    lane_t &a = ln[8 * grp + g];
                    init_pair(a.S.p, x, qlen, tlen, h0s ? h0s[ti[grp][x]] : T.h0, p->w < wl ? p->w : wl);
                    for (int s = 0; s < NS; ++s) {
                        uint32_t m[4] = {0, 0, 0, 0}, nb = 0;
                        for (int c = 0; c < 8; ++c) {
                            const int j = 64 * s + 8 * g + c;
                            // (nibbles past the query read as base 0: the packed words are zero there, words past it are not fetched)
                            const int code = (valid[grp][x] && j < qlen) ? (q[j] > 4 ? 4 : q[j]) : 0;
                            if (code >= 4) nb |= 1u << c; else m[code] |= 1u << c;
                        }
                        const uint32_t packed = m[0] | (m[1] << 8) | (m[2] << 16) | (m[3] << 24);
                        if (x == 0) { a.mA[s] = packed; a.WNs[s] = nb; } else { a.mB[s] = packed; a.WNs[s] |= nb << 16; }
                        if (nb && valid[grp][x]) nqs[s] = true;
                    }
                }
            }
        for (int l = 0; l < 64; ++l) L::init_row(ln[l].S, k, l & 7);
        std::vector<rowp> rv(64);
        for (int i = 0;; ++i) {
            bool any = false, anybite = false;
            for (int l = 0; l < 64; ++l) {
                row_begin2(ln[l].S.p, i, rv[l]);
                any = any || rv[l].ACT != 0;
                anybite = anybite || rv[l].BITE != 0;
            }
            if (!any) break;
            if (anybite) for (int l = 0; l < 64; ++l) L::zero_dropped(ln[l].S, rv[l], l & 7);
            typename L::rowk rk[64];
            for (int l = 0; l < 64; ++l) {
                int tb[2];
                for (int x = 0; x < 2; ++x) {
                    int b = half_of(rv[l].ACT, x) ? tq[l >> 3][x][i] : 0;
                    tb[x] = b > 4 ? 4 : b;
                }
                rk[l] = L::row_consts(ln[l].S.p, k, i, tb);
            }
            uint32_t Hc[64], Fc[64], mk2[64], Fnz[64], Lnz[64], hfin[64];
            for (int l = 0; l < 64; ++l) { Hc[l] = rk[l].h1init; Fc[l] = 0; mk2[l] = 0; Fnz[l] = 0xffffffffu; Lnz[l] = 0; hfin[l] = 0; }
            for (int s = 0; s < NS; ++s) {
                bool need = false;
                for (int l = 0; l < 64; ++l) need = need || L::needs_stripe(rk[l], rv[l].ACT, s) != 0;
                if (!need) continue;
                typename L::stripe_in si[64];
                uint32_t D[64], x[64], Fin[64], hl[64], fo[64], mkb[64], nz8[64];
                for (int l = 0; l < 64; ++l) {
                    const uint32_t J0d = dup16((int)L::j0_of(s, l & 7));
                    D[l] = nqs[s] ? L::template phase_a<true>(ln[l].S.T[s], rk[l], k, ln[l].mA[s], ln[l].mB[s], ln[l].WNs[s], J0d, si[l])
                                  : L::template phase_a<false>(ln[l].S.T[s], rk[l], k, ln[l].mA[s], ln[l].mB[s], ln[l].WNs[s], J0d, si[l]);
                }
                // the carry of the stripe before enters lane 0's offer; inclusive max-plus scan, decay 8 e_ins per lane
                for (int l = 0; l < 64; ++l) x[l] = (l & 7) == 0 ? pk_max(D[l], pk_subs_vs(Fc[l], E8)) : D[l];
                const uint32_t dec[3] = {E8, E16, E32};
                for (int st = 0; st < 3; ++st) {
                    uint32_t y[64];
                    const int d = 1 << st;
                    for (int l = 0; l < 64; ++l) y[l] = (l & 7) >= d ? x[l - d] : 0u;
                    for (int l = 0; l < 64; ++l) x[l] = pk_max(x[l], pk_subs_vs(y[l], dec[st]));
                }
                for (int l = 0; l < 64; ++l) Fin[l] = (l & 7) == 0 ? Fc[l] : x[l - 1];
                for (int l = 0; l < 64; ++l) {
                    fo[l] = Fin[l];
                    if (nqs[s]) L::template phase_b<true>(ln[l].S.T[s], si[l], rk[l], k, hl[l], fo[l], mkb[l], nz8[l]);
                    else L::template phase_b<false>(ln[l].S.T[s], si[l], rk[l], k, hl[l], fo[l], mkb[l], nz8[l]);
                }
                for (int l = 0; l < 64; ++l) {
                    const uint32_t Hin = (l & 7) == 0 ? Hc[l] : hl[l - 1];
                    L::phase_c(ln[l].S.T[s][0], nz8[l], si[l], k, Hin);
                    L::foldv(mk2[l], Fnz[l], Lnz[l], mkb[l], nz8[l], si[l].J0d, k);
                    hfin[l] = pk_max(hfin[l], L::hfin_cand(si[l], hl[l]));
                }
                for (int l = 0; l < 64; l += 8) { Hc[l] = hl[l + 7]; Fc[l] = fo[l + 7]; }      // (only lane 0 of a group reads them)
            }
            for (int grp = 0; grp < 8; ++grp) {
                uint32_t m = 0, L2v = 0, F = 0xffffffffu, h = 0;
                for (int g = 0; g < 8; ++g) {
                    const int l = 8 * grp + g;
                    m = pk_max(m, mk2[l]); L2v = pk_max(L2v, Lnz[l]); F = pk_min(F, Fnz[l]); h = pk_max(h, hfin[l]);
                }
                for (int g = 0; g < 8; ++g) {
                    const int l = 8 * grp + g;
                    row_tail2<SYM>(ln[l].S.p, k, i, rv[l].ACT, L::hfin_of(rk[l], h), m, F, L2v);
                }
            }
        }
        for (int grp = 0; grp < 8; ++grp)
            for (int x = 0; x < 2; ++x) {
                if (!valid[grp][x]) continue;
                const ext_out s = pair_result(ln[8 * grp].S.p, x);
                bsw_ext &e = out[ti[grp][x]];
                e.score = s.mx; e.qle = s.max_j + 1; e.tle = s.max_i + 1; e.gtle = s.max_ie + 1;
                e.gscore = s.gscore; e.max_off = s.max_off; e.aw = p->w; e.cells = s.cells;
            }
    }
};

// ns = stripes of 64 columns (3: the 8-bit classes up to 136 columns, 4: the 232-column class)
extern "C" int lane2g_model_run(const bsw_params *p, const bsw_task *tasks, int side, const uint32_t *order, size_t n,
                                const int32_t *h0s, bsw_ext *out, int ns)
{
    if (!p || !tasks || !order || !out) return -1;
    if (p->mat[1] > 0 || p->mat[24] > 0 || -p->mat[1] < -p->mat[24]) return -2;
    if (p->o_del + p->e_del > 255 || p->o_ins + p->e_ins > 255 || p->mat[0] - p->mat[1] > 255) return -2;
    const bool sym = p->o_del == p->o_ins && p->e_del == p->e_ins, vm = p->variant == BSW_VARIANT_M;
    for (size_t w0 = 0; w0 < n; w0 += 16) {
#define RUNG(NS) do { \
        if (!vm && sym) group_model<NS, false, true>::run(p, tasks, side, order, n, w0, h0s, out); \
        else if (!vm) group_model<NS, false, false>::run(p, tasks, side, order, n, w0, h0s, out); \
        else if (sym) group_model<NS, true, true>::run(p, tasks, side, order, n, w0, h0s, out); \
        else group_model<NS, true, false>::run(p, tasks, side, order, n, w0, h0s, out); } while (0)
        if (ns == 3) RUNG(3); else if (ns == 4) RUNG(4); else return -3;
#undef RUNG
    }
    return 0;
}
