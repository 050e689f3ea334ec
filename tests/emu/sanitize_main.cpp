// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
// Sanitizer driver for the host-emulated kernel build: calls every compute entry point of include/mentflow_hip.h on
// small synthetic inputs.  Built by tests/emu/build_sanitize.sh with -fsanitize=address,undefined (the emulator
// annotates its fiber switches for ASan and gives every workgroup an exactly-sized, guard-paged dynamic LDS block), so
// any out-of-range LDS / global index, use of uninitialised stack slots through a pointer, or undefined arithmetic in
// flow.hip / kde.hip stops the program.   `--provoke-lds-overflow`: self-test of the guard (must crash).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/mentflow_hip.h"

static unsigned g_seed = 12345u;
static float frand() {
    g_seed = g_seed * 1664525u + 1013904223u;
    return (float)((g_seed >> 8) & 0xffff) / 65536.0f - 0.5f;
}
static std::vector<float> rnd(size_t n, float scale) {
    std::vector<float> v(n);
    for (auto& x : v) x = scale * 2.0f * frand();
    return v;
}
#define CK(call)                                                                 \
    do {                                                                         \
        if ((call) != 0) {                                                       \
            fprintf(stderr, "FAILED %s: %s\n", #call, mf_last_error());          \
            exit(2);                                                             \
        }                                                                        \
    } while (0)

static void all_finite(const std::vector<float>& v, const char* what) {
    for (float x : v)
        if (!std::isfinite(x)) {
            fprintf(stderr, "non-finite value in %s\n", what);
            exit(3);
        }
}

static void flow_rqs(int d, int L, int K, int64_t n, bool fused) {
    setenv("MENTFLOW_BWD_FUSED", fused ? "1" : "0", 1);
    const int64_t F = mf_flow_image_floats(d, L);
    std::vector<float> image = rnd(F, 0.3f);
    std::vector<int32_t> order(d);
    for (int i = 0; i < d; ++i) order[i] = d - 1 - i;
    std::vector<float> x = rnd(n * d, 2.0f), y(n * d), logp(n), x2(n * d);
    x[0] = 7.0f;                                   // identity tail of the spline
    CK(mf_flow_rqs_layer_fwd(image.data(), d, L, K, order.data(), x.data(), n, y.data(), nullptr, logp.data(), 1, nullptr));
    CK(mf_flow_rqs_layer_fwd(image.data(), d, L, K, nullptr, x.data(), n, y.data(), logp.data(), logp.data(), 0, nullptr));
    all_finite(y, "rqs y");
    all_finite(logp, "rqs logp");
    std::vector<float> gy = rnd(n * d, 1.0f), gl = rnd(n, 1.0f), gx(n * d);
    const int rows = mf_flow_bwd_slab_rows(n, d, L, order.data());
    const int64_t sf = mf_flow_bwd_scratch_floats(n, d, L, order.data());
    std::vector<float> slab((size_t)rows * F, NAN), scratch((size_t)(sf > 0 ? sf : 1));
    CK(mf_flow_rqs_layer_bwd(image.data(), d, L, K, order.data(), x.data(), n, gy.data(), gl.data(), gx.data(), slab.data(), rows,
                             0, scratch.data(), (int64_t)scratch.size(), nullptr));
    CK(mf_flow_rqs_layer_bwd(image.data(), d, L, K, order.data(), x.data(), n, gy.data(), gl.data(), nullptr, slab.data(), rows,
                             1, scratch.data(), (int64_t)scratch.size(), nullptr));
    all_finite(gx, "rqs gx");
    // activation hand-off (ABI 4): forward that saves, backward that loads; exactly-sized buffer (ASan guards its end), and the
    // parameter-gradient slab must come out bit for bit as the recompute backward wrote it
    const int lvl_max = mf_flow_rqs_act_level(d, L, K, order.data());
    for (int level = 1; level <= lvl_max; ++level) {
        const int64_t af = mf_flow_rqs_act_floats(n, d, L, K, level);
        std::vector<float> act((size_t)af, NAN), y2(n * d), logp2(n), gx2(n * d), slab2((size_t)rows * F, NAN);
        CK(mf_flow_rqs_layer_fwd_save(image.data(), d, L, K, order.data(), x.data(), n, y2.data(), nullptr, logp2.data(), 1,
                                      act.data(), af, level, nullptr));
        all_finite(act, "rqs act");
        CK(mf_flow_rqs_layer_bwd_saved(image.data(), d, L, K, order.data(), x.data(), n, gy.data(), gl.data(), gx2.data(),
                                       slab2.data(), rows, 0, act.data(), af, level, nullptr));
        std::vector<float> slab1((size_t)rows * F, NAN);
        CK(mf_flow_rqs_layer_bwd(image.data(), d, L, K, order.data(), x.data(), n, gy.data(), gl.data(), gx.data(), slab1.data(),
                                 rows, 0, scratch.data(), (int64_t)scratch.size(), nullptr));
        if (memcmp(gx.data(), gx2.data(), gx.size() * sizeof(float)) != 0 ||
            memcmp(slab1.data(), slab2.data(), slab1.size() * sizeof(float)) != 0) {
            fprintf(stderr, "activation hand-off level %d differs from the recompute backward (d=%d L=%d K=%d)\n", level, d, L, K);
            exit(4);
        }
    }
    // reduce through an identity index over the first layer's image: every slot a parameter can map to must have been
    // written by every workgroup (NaN-initialised slab: an unwritten slot shows up here) — restricted to the slots the
    // packing really uses is the Python side's job; here: finite where written
    std::vector<int32_t> gidx(F);
    for (int64_t i = 0; i < F; ++i) gidx[i] = -1;
    std::vector<float> gflat(F);
    CK(mf_flow_grad_reduce(slab.data(), 1, rows, F, gidx.data(), gflat.data(), F, nullptr));
    CK(mf_flow_rqs_layer_inv(image.data(), d, L, K, order.data(), y.data(), n, x2.data(), nullptr));
    printf("  rqs d=%d L=%d K=%d n=%ld %s: slab rows %d, scratch %ld floats\n", d, L, K, (long)n, fused ? "fused" : "two-kernel", rows, (long)sf);
}

static void flow_affine(int d, int L, int64_t n, bool fused) {
    setenv("MENTFLOW_BWD_FUSED", fused ? "1" : "0", 1);
    const int64_t F = mf_flow_affine_image_floats(d, L);
    std::vector<float> image = rnd(F, 0.3f);
    std::vector<int32_t> order(d);
    for (int i = 0; i < d; ++i) order[i] = i;
    std::vector<float> x = rnd(n * d, 2.0f), y(n * d), logp(n), x2(n * d);
    CK(mf_flow_affine_layer_fwd(image.data(), d, L, order.data(), x.data(), n, y.data(), nullptr, logp.data(), 1, nullptr));
    std::vector<float> gy = rnd(n * d, 1.0f), gl = rnd(n, 1.0f), gx(n * d);
    const int rows = mf_flow_affine_bwd_slab_rows(n);
    const int64_t sf = mf_flow_affine_bwd_scratch_floats(n, L);
    std::vector<float> slab((size_t)rows * F), scratch((size_t)(sf > 0 ? sf : 1));
    CK(mf_flow_affine_layer_bwd(image.data(), d, L, order.data(), x.data(), n, gy.data(), gl.data(), gx.data(), slab.data(), rows, 0,
                                scratch.data(), (int64_t)scratch.size(), nullptr));
    all_finite(gx, "affine gx");
    CK(mf_flow_affine_layer_inv(image.data(), d, L, order.data(), y.data(), n, x2.data(), nullptr));
    printf("  affine d=%d L=%d n=%ld %s\n", d, L, (long)n, fused ? "fused" : "two-kernel");
}

// wide conditioner family (ABI 5): exactly-sized image / scratch / slab buffers, so that a fragment block, a scratch tile or a slab
// position addressed out of range runs into ASan's redzone
static void flow_wide(int d, int hidden, int L, int K, int64_t n) {
    const int nblk = K ? d : 1;
    const int64_t F = mf_flow_wide_image_floats(L, nblk), G = mf_flow_wide_grad_floats(L, nblk);
    std::vector<float> image = rnd(F, 0.3f);
    std::vector<int32_t> order(d);
    for (int i = 0; i < d; ++i) order[i] = d - 1 - i;
    std::vector<float> x = rnd(n * d, 2.0f), y(n * d), logp(n), x2(n * d);
    x[0] = 7.0f;
    CK(mf_flow_wide_layer_fwd(image.data(), d, hidden, L, K, order.data(), x.data(), n, y.data(), nullptr, logp.data(), 1, nullptr));
    CK(mf_flow_wide_layer_fwd(image.data(), d, hidden, L, K, nullptr, x.data(), n, y.data(), logp.data(), logp.data(), 0, nullptr));
    all_finite(y, "wide y");
    all_finite(logp, "wide logp");
    std::vector<float> gy = rnd(n * d, 1.0f), gl = rnd(n, 1.0f), gx(n * d);
    const int rows = mf_flow_wide_bwd_slab_rows(n);
    const int64_t sf = mf_flow_wide_bwd_scratch_floats(n, d, L, K);
    std::vector<float> slab((size_t)rows * G, NAN), scratch((size_t)sf, NAN);
    CK(mf_flow_wide_layer_bwd(image.data(), d, hidden, L, K, order.data(), x.data(), n, gy.data(), gl.data(), gx.data(), slab.data(),
                              rows, 0, scratch.data(), sf, nullptr));
    all_finite(scratch, "wide scratch");                   // every tile of the scratch was written
    CK(mf_flow_wide_layer_bwd(image.data(), d, hidden, L, K, order.data(), x.data(), n, gy.data(), gl.data(), nullptr, slab.data(),
                              rows, 1, scratch.data(), sf, nullptr));
    all_finite(gx, "wide gx");
    CK(mf_flow_wide_layer_inv(image.data(), d, hidden, L, K, order.data(), y.data(), n, x2.data(), nullptr));
    all_finite(x2, "wide inverse");
    // activation hand-off: exactly-sized buffer, every float of it written by the forward; the backward that loads it must reproduce
    // dL/dx and the slab of the recomputing backward bit for bit
    {
        const int64_t af = mf_flow_wide_act_floats(n, d, L, K);
        std::vector<float> act((size_t)af, NAN), y2(n * d), logp2(n), gx1(n * d), gx2(n * d), slab1((size_t)rows * G, NAN),
            slab2((size_t)rows * G, NAN), scratch2((size_t)sf, NAN);
        CK(mf_flow_wide_layer_fwd_save(image.data(), d, hidden, L, K, order.data(), x.data(), n, y2.data(), nullptr, logp2.data(), 1,
                                       act.data(), af, nullptr));
        all_finite(act, "wide act");
        CK(mf_flow_wide_layer_bwd(image.data(), d, hidden, L, K, order.data(), x.data(), n, gy.data(), gl.data(), gx1.data(), slab1.data(),
                                  rows, 0, scratch.data(), sf, nullptr));
        CK(mf_flow_wide_layer_bwd_saved(image.data(), d, hidden, L, K, order.data(), x.data(), n, gy.data(), gl.data(), gx2.data(),
                                        slab2.data(), rows, 0, scratch2.data(), sf, act.data(), af, nullptr));
        if (memcmp(gx1.data(), gx2.data(), gx1.size() * sizeof(float)) != 0) {
            fprintf(stderr, "wide hand-off: dL/dx differs from the recomputing backward (d=%d hidden=%d L=%d bins=%d)\n", d, hidden, L, K);
            exit(4);
        }
        for (size_t i = 0; i < slab1.size(); ++i)            // positions no job writes stay NaN in both
            if (memcmp(&slab1[i], &slab2[i], sizeof(float)) != 0) {
                fprintf(stderr, "wide hand-off: slab position %zu differs (d=%d hidden=%d L=%d bins=%d)\n", i, d, hidden, L, K);
                exit(4);
            }
    }
    printf("  wide d=%d hidden=%d L=%d bins=%d n=%ld: image %ld, gradient image %ld floats, slab rows %d, scratch %ld floats\n", d, hidden,
           L, K, (long)n, (long)F, (long)G, rows, (long)sf);
}

static std::vector<float> centres(int B, float lo, float hi, std::vector<float>* edges = nullptr) {
    std::vector<float> e(B + 1), c(B);
    for (int i = 0; i <= B; ++i) e[i] = lo + (hi - lo) * (float)i / (float)B;
    for (int i = 0; i < B; ++i) c[i] = 0.5f * (e[i] + e[i + 1]);
    if (edges) *edges = e;
    return c;
}

static void kde1d(int64_t n, int d, int P, int B, float bw, int radius) {
    std::vector<float> edges, c = centres(B, -3.5f, 3.5f, &edges);
    std::vector<float> x = rnd(n * d, 2.5f), V = rnd((size_t)P * d, 0.6f), S((size_t)P * B), gS = rnd((size_t)P * B, 1.0f), gx(n * d);
    x[3] = INFINITY; x[d + 1] = NAN; x[2 * d] = 1e30f; x[3 * d + 2] = -INFINITY;
    const float sigma = bw * (c[1] - c[0]);
    std::vector<unsigned char> ws((size_t)mf_proj_kde_ws_bytes(P, B));
    CK(mf_proj_kde1d_fwd(x.data(), n, d, V.data(), P, c.data(), B, sigma, radius, S.data(), ws.data(), nullptr));
    all_finite(S, "kde1d S");
    CK(mf_proj_kde1d_bwd(x.data(), n, d, V.data(), P, c.data(), B, sigma, radius, gS.data(), gx.data(), 0, nullptr));
    CK(mf_proj_kde1d_bwd(x.data(), n, d, V.data(), P, c.data(), B, sigma, radius, gS.data(), gx.data(), 1, nullptr));
    all_finite(gx, "kde1d gx");
    std::vector<int32_t> counts((size_t)P * B);
    CK(mf_proj_hist1d_counts(x.data(), n, d, V.data(), P, edges.data(), B, counts.data(), nullptr));
    // tail: normalisation + the three discrepancies, forward and adjoint
    std::vector<float> meas = S, ghat((size_t)P * B), D(P), gD = rnd(P, 1.0f), gS2((size_t)P * B);
    for (int kind = 0; kind < 3; ++kind) {
        CK(mf_hist_norm_discrepancy_fwd(S.data(), P, B, 1, 1.0f / (float)n, c[1] - c[0], 1e-10f, meas.data(), kind, 1e-12f, (float)B,
                                        ghat.data(), D.data(), nullptr));
        CK(mf_hist_norm_discrepancy_bwd(S.data(), P, B, 1, 1.0f / (float)n, c[1] - c[0], 1e-10f, meas.data(), kind, 1e-12f, (float)B,
                                        gD.data(), nullptr, gS2.data(), nullptr));
    }
    printf("  kde1d n=%ld d=%d P=%d B=%d radius=%d\n", (long)n, d, P, B, radius);
}

static void kde2d(int64_t n, int d, int P, int Bx, int By, float bwx, float bwy, int rx, int ry) {
    std::vector<float> ex, ey, cx = centres(Bx, -3.0f, 3.0f, &ex), cy = centres(By, -2.5f, 2.5f, &ey);
    std::vector<float> x = rnd(n * d, 2.2f), V0 = rnd((size_t)P * d, 0.6f), V1 = rnd((size_t)P * d, 0.6f);
    x[1] = NAN; x[d] = INFINITY; x[2 * d + 1] = -1e30f;
    std::vector<float> S((size_t)P * Bx * By), gS = rnd((size_t)P * Bx * By, 1.0f), gx(n * d);
    std::vector<unsigned char> ws((size_t)mf_proj_kde_ws_bytes(P, Bx * By));
    const float sx = bwx * (cx[1] - cx[0]), sy = bwy * (cy[1] - cy[0]);
    CK(mf_proj_kde2d_fwd(x.data(), n, d, V0.data(), V1.data(), P, cx.data(), Bx, sx, rx, cy.data(), By, sy, ry, S.data(), ws.data(), nullptr));
    all_finite(S, "kde2d S");
    CK(mf_proj_kde2d_bwd(x.data(), n, d, V0.data(), V1.data(), P, cx.data(), Bx, sx, rx, cy.data(), By, sy, ry, gS.data(), gx.data(), 0, nullptr));
    all_finite(gx, "kde2d gx");
    std::vector<int32_t> counts((size_t)P * Bx * By);
    CK(mf_proj_hist2d_counts(x.data(), n, d, V0.data(), V1.data(), P, ex.data(), Bx, ey.data(), By, counts.data(), nullptr));
    printf("  kde2d n=%ld d=%d P=%d %dx%d radii %d,%d\n", (long)n, d, P, Bx, By, rx, ry);
}

static void tail(int64_t n, int d) {
    std::vector<float> x = rnd(n * d, 1.0f), logp = rnd(n, 1.0f), out(2), gx(n * d), coef(1, 0.5f), u(n * d);
    std::vector<double> acc(MF_ENTROPY_SCRATCH_DOUBLES);
    CK(mf_mc_entropy_sums(x.data(), logp.data(), n, d, out.data(), acc.data(), nullptr));
    CK(mf_scale_rows(x.data(), n, d, coef.data(), 2.0f, gx.data(), 0, nullptr));
    std::vector<int32_t> idx(n);
    for (int64_t i = 0; i < n; ++i) idx[i] = (i % 3 == 0) ? -1 : (int32_t)((i * 7) % (n * d));
    std::vector<float> dst(n);
    CK(mf_gather_f32(x.data(), idx.data(), dst.data(), n, 0, nullptr));
    if (d == 2 || d >= 4) {
        for (int order = 3; order <= 5; ++order) {
            CK(mf_multipole_kick_fwd(x.data(), n, d, order, 0.7f, order & 1, u.data(), nullptr));
            CK(mf_multipole_kick_bwd(x.data(), n, d, order, 0.7f, order & 1, u.data(), gx.data(), nullptr));
        }
    }
    printf("  tail n=%ld d=%d\n", (long)n, d);
}

#ifdef MF_EMU
static void provoke_lds_overflow() {
    // a "kernel" that writes ONE float past its dynamic LDS block: the guard page / ASan poison must stop it
    emu::launch(dim3(1), dim3(64), 1024 * sizeof(float), [&]() {
        float* lds = reinterpret_cast<float*>(((uintptr_t)emu::g_block->dyn_smem + 63) & ~(uintptr_t)63);
        lds[threadIdx.x] = 1.0f;
        __syncthreads();
        if (threadIdx.x == 0) lds[1024 + 16] = 2.0f;       // 64 bytes beyond the end (beyond any alignment slack)
    });
    printf("LDS overflow was NOT caught\n");
}
#endif

int main(int argc, char** argv) {
    if (argc > 1 && strcmp(argv[1], "--provoke-lds-overflow") == 0) {
#ifdef MF_EMU
        provoke_lds_overflow();
#endif
        return 0;
    }
    if (mf_is_emulation() != 1) {
        fprintf(stderr, "this driver is for the host-emulated build only\n");
        return 4;
    }
    printf("flow kernels\n");
    flow_rqs(6, 3, 20, 300, true);       // fused: three 4-tile groups, ragged last tile
    flow_rqs(6, 3, 20, 70, false);       // two-kernel + outer_accum
    flow_rqs(2, 2, 8, 150, true);
    flow_rqs(7, 3, 20, 40, true);        // d = 7 does not fit the fused kernel: two-kernel path whatever the switch
    flow_rqs(5, 3, 8, 33, false);
    flow_affine(2, 3, 200, true);
    flow_affine(7, 2, 45, false);
    flow_wide(6, 128, 3, 20, 150);       // four hidden tiles, ragged last particle tile
    flow_wide(9, 70, 2, 8, 70);          // three tiles in use, two input-layer k-groups
    flow_wide(3, 128, 4, 13, 40);        // run-time bins, four hidden layers
    flow_wide(16, 100, 1, 0, 90);        // affine, widest input
    printf("KDE kernels\n");
    kde1d(1500, 6, 7, 64, 0.5f, 4);      // factorised radius-4 window
    kde1d(700, 3, 5, 85, 0.6f, 5);       // runtime radius
    kde1d(300, 2, 100, 16, 0.5f, 4);     // many projections, several groups
    kde2d(700, 6, 3, 20, 24, 0.5f, 0.5f, 4, 4);
    kde2d(400, 4, 2, 85, 85, 0.5f, 0.45f, 4, 4);
    kde2d(300, 2, 2, 17, 13, 0.6f, 0.3f, 5, 3);
    tail(1000, 6);
    tail(333, 2);
    printf("SANITIZE OK\n");
    return 0;
}
