// v1t_amd - the per-mouse "tails" of a training step as ONE launch per stage (C-ABI: v1t_tails_*, include/v1t_amd.h).
// The reference's step loops over mice (train.py:97-111): per mouse a core shifter (core_shifter.py:24-40), the readout's sample positions
// (gaussian2d.py:188-235), the Gaussian2d readout (:270-276), ELU + 1 and the Poisson loss (models/utils.py:109-118, losses.py:141-166) and
// their backward. The core is shared and runs once over all local mice; what is left per mouse are ~11 small
// latency-bound kernels. Here every stage takes a table of units (one per local mouse-batch) and runs them in one launch: the kernels are the
// single-unit ones (readout.hip, gridprep.hip, elementwise.hip), a workgroup finds its unit from prefix sums in the kernel arguments.
#include <cmath>
#include <vector>
#include "readout.h"
#include "gridprep.h"
#include "elementwise.h"
#include "../../include/v1t_amd.h"

namespace {
struct Geo { const float* z; float* dz; long long zsb, zsc; int C, gh, gw; };

ReadoutArgs readout_args(const v1t_tail_unit& u, const Geo& g) {
    ReadoutArgs a{};
    a.z = g.z ? g.z + (size_t)u.image_offset * g.zsb : nullptr;
    a.zsb = g.zsb; a.zsc = g.zsc; a.B = u.n_images; a.C = g.C; a.H = g.gh; a.W = g.gw; a.N = u.n_neurons;
    a.grid = u.grid; a.feat = u.feat; a.FS = u.feat_stride; a.bias = u.bias; a.out = u.u;
    a.gout = u.du; a.dz = g.dz ? g.dz + (size_t)u.image_offset * g.zsb : nullptr; a.dzsb = g.zsb; a.dzsc = g.zsc;
    a.dgrid = u.dgrid; a.dfeat = u.dfeat; a.dbias = u.dbias;
    return a;
}
GridArgs grid_args(const v1t_tail_unit& u) {
    GridArgs a{};
    a.B = u.n_images; a.N = u.n_neurons; a.gd = u.grid_dim; a.src = u.src; a.W0 = u.gp[0]; a.b0 = u.gp[1]; a.W2 = u.gp[2]; a.b2 = u.gp[3];
    a.mu_free = u.mu; a.sigma = u.sigma; a.eps = u.eps; a.shift = u.shift; a.grid = u.grid;
    a.dgrid = u.dgrid; a.dW0 = u.dgp[0]; a.db0 = u.dgp[1]; a.dW2 = u.dgp[2]; a.db2 = u.dgp[3]; a.dmu_free = u.dmu; a.dsigma = u.dsigma;
    a.dshift = u.shift ? u.dshift : nullptr;
    return a;
}
ShifterArgs shifter_args(const v1t_tail_unit& u) {
    ShifterArgs a{};
    a.B = u.n_images; a.pupil = u.pupil;
    a.W0 = u.sp[0]; a.b0 = u.sp[1]; a.W2 = u.sp[2]; a.b2 = u.sp[3]; a.W4 = u.sp[4]; a.b4 = u.sp[5];
    a.shift = u.shift; a.dshift = u.dshift;
    a.dW0 = u.dsp[0]; a.db0 = u.dsp[1]; a.dW2 = u.dsp[2]; a.db2 = u.dsp[3]; a.dW4 = u.dsp[4]; a.db4 = u.dsp[5];
    return a;
}
bool bad_unit(const v1t_tail_unit& u) {
    if (u.n_images <= 0 || u.n_neurons <= 0 || u.image_offset < 0 || !u.grid || !u.sigma || !u.feat || !u.eps) return true;
    if (u.grid_dim > 0 ? (!u.src || !u.gp[0] || !u.gp[1] || !u.gp[2] || !u.gp[3]) : !u.mu) return true;
    if (u.shift) for (int i = 0; i < 6; ++i) if (!u.sp[i]) return true;
    return false;
}
}  // namespace

// chunks of <= TAILS_MAX_UNITS units per launch (a step has 7 mice; more only in synthetic set-ups)
#define FOR_CHUNKS(n_units) for (int c0 = 0; c0 < (n_units); c0 += TAILS_MAX_UNITS) if (const int cn = ((n_units) - c0 < TAILS_MAX_UNITS ? (n_units) - c0 : TAILS_MAX_UNITS); cn > 0)

extern "C" int v1t_tails_prepare(const v1t_tail_unit* units, int n_units, unsigned long long eps_seed, int gh, int gw, void* stream) {
    if (!units || n_units < 0 || gh <= 0 || gw <= 0) return V1T_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < n_units; ++i) if (bad_unit(units[i]) || !units[i].rws) return V1T_ERR_ARG;
    FOR_CHUNKS(n_units) {
        const v1t_tail_unit* u = units + c0;
        // core shifter forward (units with a shifter)
        ShifterArgs sh[TAILS_MAX_UNITS];
        int nsh = 0;
        for (int i = 0; i < cn; ++i) if (u[i].shift) { if (!u[i].pupil) return V1T_ERR_ARG; sh[nsh++] = shifter_args(u[i]); }
        int rc = launch_shifter_multi(sh, nsh, false, s);
        if (rc) return rc;
        // position noise (units that do not replay an injected one)
        float* eo[TAILS_MAX_UNITS]; long long en[TAILS_MAX_UNITS]; uint32_t es[TAILS_MAX_UNITS];
        int ne = 0;
        for (int i = 0; i < cn; ++i) if (u[i].fill_eps) { eo[ne] = u[i].eps; en[ne] = 2LL * u[i].n_images * u[i].n_neurons; es[ne] = u[i].eps_stream; ++ne; }
        rc = launch_normal_fill_multi(eo, en, es, ne, eps_seed, s);
        if (rc) return rc;
        // sample positions, then the counting sort of the taps (needs the positions only)
        GridArgs ga[TAILS_MAX_UNITS];
        ReadoutArgs ra[TAILS_MAX_UNITS];
        void* ws[TAILS_MAX_UNITS]; size_t wsb[TAILS_MAX_UNITS];
        const Geo g{nullptr, nullptr, 0, 0, 1, gh, gw};
        for (int i = 0; i < cn; ++i) { ga[i] = grid_args(u[i]); ra[i] = readout_args(u[i], g); ws[i] = u[i].rws; wsb[i] = (size_t)u[i].rws_bytes; }
        rc = launch_grid_fwd_multi(ga, cn, s);
        if (rc) return rc;
        rc = launch_readout_multi(ra, ws, wsb, cn, TAILS_SORT, s);
        if (rc) return rc;
    }
    return V1T_OK;
}

extern "C" int v1t_tails_forward(const v1t_tail_unit* units, int n_units, const float* tokens, float* dtokens, long long zsb, long long zsc, int C, int gh,
                                 int gw, float* loss_total, void* stream) {
    if (!units || n_units < 0 || !tokens || !dtokens || C <= 0 || gh <= 0 || gw <= 0) return V1T_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const Geo g{tokens, dtokens, zsb, zsc, C, gh, gw};
    for (int i = 0; i < n_units; ++i) if (bad_unit(units[i]) || !units[i].u || !units[i].du || !units[i].response || !units[i].loss || !units[i].rws) return V1T_ERR_ARG;
    FOR_CHUNKS(n_units) {
        const v1t_tail_unit* u = units + c0;
        ReadoutArgs ra[TAILS_MAX_UNITS];
        LossArgs la[TAILS_MAX_UNITS];
        void* ws[TAILS_MAX_UNITS]; size_t wsb[TAILS_MAX_UNITS];
        for (int i = 0; i < cn; ++i) {
            ra[i] = readout_args(u[i], g);
            ws[i] = u[i].rws; wsb[i] = (size_t)u[i].rws_bytes;
            LossArgs l{};
            l.u = u[i].u; l.y = u[i].response; l.yhat = u[i].yhat; l.du = u[i].du; l.loss = u[i].loss; l.loss_total = loss_total;
            l.n = (long long)u[i].n_images * u[i].n_neurons; l.loss_scale = u[i].loss_scale; l.gscale = 1.0f;
            la[i] = l;
        }
        int rc = launch_readout_multi(ra, ws, wsb, cn, TAILS_FWD, s);
        if (rc) return rc;
        rc = launch_elu1_poisson_multi(la, cn, s);
        if (rc) return rc;
        rc = launch_readout_multi(ra, ws, wsb, cn, TAILS_DZ, s);
        if (rc) return rc;
    }
    return V1T_OK;
}

extern "C" int v1t_tails_backward(const v1t_tail_unit* units, int n_units, const float* tokens, long long zsb, long long zsc, int C, int gh, int gw,
                                  void* stream) {
    if (!units || n_units < 0 || !tokens || C <= 0 || gh <= 0 || gw <= 0) return V1T_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const Geo g{tokens, nullptr, zsb, zsc, C, gh, gw};
    for (int i = 0; i < n_units; ++i) if (bad_unit(units[i]) || !units[i].du || !units[i].dgrid || !units[i].dfeat || !units[i].gws) return V1T_ERR_ARG;
    FOR_CHUNKS(n_units) {
        const v1t_tail_unit* u = units + c0;
        ReadoutArgs ra[TAILS_MAX_UNITS];
        GridArgs ga[TAILS_MAX_UNITS];
        ShifterArgs sh[TAILS_MAX_UNITS];
        void* ws[TAILS_MAX_UNITS]; size_t wsb[TAILS_MAX_UNITS];
        void* gws[TAILS_MAX_UNITS]; size_t gwsb[TAILS_MAX_UNITS];
        int nsh = 0;
        for (int i = 0; i < cn; ++i) {
            ra[i] = readout_args(u[i], g);
            ga[i] = grid_args(u[i]);
            ws[i] = u[i].rws; wsb[i] = (size_t)u[i].rws_bytes;
            gws[i] = u[i].gws; gwsb[i] = (size_t)u[i].gws_bytes;
            if (u[i].shift) {
                for (int k = 0; k < 6; ++k) if (!u[i].dsp[k]) return V1T_ERR_ARG;
                if (!u[i].dshift || !u[i].pupil) return V1T_ERR_ARG;
                sh[nsh++] = shifter_args(u[i]);
            }
        }
        int rc = launch_readout_multi(ra, ws, wsb, cn, TAILS_PARAMS, s);  // d grid, d features, d bias
        if (rc) return rc;
        rc = launch_grid_bwd_multi(ga, gws, gwsb, cn, s);                  // predictor / mu / sigma gradients, d shift
        if (rc) return rc;
        rc = launch_shifter_multi(sh, nsh, true, s);
        if (rc) return rc;
    }
    return V1T_OK;
}

extern "C" int v1t_adamw_multi(const v1t_adam_range* r, int n, float beta1, float beta2, float eps, float weight_decay, int zero_grad, void* stream) {
    if (!r || n < 0) return V1T_ERR_ARG;
    std::vector<AdamArgs> a((size_t)n);
    for (int i = 0; i < n; ++i) {
        if (!r[i].p || !r[i].g || !r[i].m || !r[i].v || r[i].step < 1) return V1T_ERR_ARG;
        AdamArgs& x = a[i];
        x = AdamArgs{};
        x.p = r[i].p; x.g = r[i].g; x.m = r[i].m; x.v = r[i].v; x.n = r[i].n; x.lr = r[i].lr; x.beta1 = beta1; x.beta2 = beta2; x.eps = eps;
        x.weight_decay = weight_decay; x.l1 = r[i].l1; x.zero_grad = zero_grad;
        x.bc1 = (float)(1.0 - std::pow((double)beta1, r[i].step));
        x.bc2 = (float)(1.0 - std::pow((double)beta2, r[i].step));
    }
    return launch_adamw_multi(a.data(), n, (hipStream_t)stream);
}

extern "C" int v1t_fill_zero(void* p, long long bytes, void* stream) {
    if (bytes < 0 || (bytes && !p) || ((uintptr_t)p & 15)) return V1T_ERR_ARG;
    return launch_fill_zero(p, bytes, (hipStream_t)stream);
}

extern "C" int v1t_inputs_multi(const float* const* images, const float* const* behaviors, const float* const* pupil_centers, const int* n_images, int n_units,
                                int C, int IH, int IW, float* out, int OH, int OW, float* beh_out, int na, int nb, void* stream) {
    if (!images || !n_images || n_units < 0 || !out || C <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0 || na < 0 || nb < 0) return V1T_ERR_ARG;
    if (beh_out && ((na && !behaviors) || (nb && !pupil_centers))) return V1T_ERR_ARG;
    for (int i = 0; i < n_units; ++i) {
        if (!images[i] || n_images[i] < 0) return V1T_ERR_ARG;
        if (beh_out && ((na && !behaviors[i]) || (nb && !pupil_centers[i]))) return V1T_ERR_ARG;
    }
    return launch_inputs_multi(images, behaviors, pupil_centers, n_images, n_units, C, IH, IW, out, OH, OW, beh_out, na, nb, (hipStream_t)stream);
}
