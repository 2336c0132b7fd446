// head.hip — the tower head of the multi-tower models in ONE forward and ONE backward launch.
//
// Reference: BaseModel.tower_forward (model/layer.py:48-56) after the towers' hidden layers — per tower the output
// Linear(->1) of its MultiLayerPerceptron (model/layer.py:193), `y_logits += other` for every other logit (the wide term
// FeaturesLinear model/layer.py:122-126, optionally the attention logit), Sigmoid, concatenation to [B, n_tower] — and, in
// the training step, BCELoss(mean) on the row's own tower (run.py:484,723) with its gradient.
//
// Before: rowdot(wide) + rowdot(towers) forward; rowdot_bwd(towers, fused BCE) + fan-in add_n + rowdot_bwd(wide) + two ordered
// reductions backward — seven launches (40 us of the C2 step) for 4096 x 4 dot products.  Here one wave handles a row: the wide
// dot product is formed once and shared by the row's towers; the backward forms every tower's logit gradient, the wide term's
// gradient (their sum), all input gradients and the per-part weight-gradient partial sums in the same pass.  Cross-row
// reductions stay order-fixed: CDC_ROWDOT_PARTS row parts, LDS accumulators per wave, waves then parts added in index order.
#include "common.h"

#define HEAD_THREADS 256
#define HEAD_WAVES (HEAD_THREADS / 64)

#define HEAD_KJ 8            /* a lane's share of a row up to K = 512 is fetched in one go (all loads in flight), longer rows loop */
__device__ __forceinline__ float head_dot(const float* __restrict__ x, const float* __restrict__ w, int K, int lane) {
    float acc = 0.f;
    if (K <= 64 * HEAD_KJ) {
        float xv[HEAD_KJ], wv[HEAD_KJ];
#pragma unroll
        for (int j = 0; j < HEAD_KJ; ++j) {
            const int k = lane + 64 * j;
            xv[j] = k < K ? x[k] : 0.f;
            wv[j] = k < K ? w[k] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < HEAD_KJ; ++j)
            if (lane + 64 * j < K) acc += xv[j] * wv[j];                 // ascending k, as the loop below
    } else {
        for (int k = lane; k < K; k += 64) acc += x[k] * w[k];
    }
    return wave_sum(acc);
}

#define HEAD_TJ 1            /* a lane's share of a tower row up to K = 64 */
__global__ void __launch_bounds__(HEAD_THREADS) k_head_fwd(const cdc_head_args a) {
    CDC_PRIO_MAIN();
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * HEAD_WAVES + (threadIdx.x >> 6);
    if (r >= a.M) return;
    bool small = !a.wide_x || a.wide_K <= 64 * HEAD_KJ;
    for (int t = 0; t < a.n_tower; ++t) small = small && a.t[t].K <= 64 * HEAD_TJ;
    if (small) {
        // every load of the row (wide term + all towers) is issued before the first reduction: one round trip, not n_tower + 1
        float wx[HEAD_KJ], ww[HEAD_KJ], tx[CDC_HEAD_MAX_TOWERS][HEAD_TJ], tw[CDC_HEAD_MAX_TOWERS][HEAD_TJ];
#pragma unroll
        for (int j = 0; j < HEAD_KJ; ++j) {
            const int k = lane + 64 * j;
            const bool in = a.wide_x && k < a.wide_K;
            wx[j] = in ? a.wide_x[r * a.ld_wide + k] : 0.f;
            ww[j] = in ? a.wide_w[k] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < CDC_HEAD_MAX_TOWERS; ++t)
#pragma unroll
            for (int j = 0; j < HEAD_TJ; ++j) {
                const int k = lane + 64 * j;
                const bool in = t < a.n_tower && k < a.t[t].K;
                tx[t][j] = in ? a.t[t].x[r * a.t[t].ldx + k] : 0.f;
                tw[t][j] = in ? a.t[t].w[k] : 0.f;
            }
        float shared = 0.f;
        if (a.wide_x) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < HEAD_KJ; ++j)
                if (lane + 64 * j < a.wide_K) acc += wx[j] * ww[j];       // ascending k per lane, then the wave sum: head_dot's order
            shared = wave_sum(acc);
            if (a.wide_bias) shared += a.wide_bias[0];
            if (lane == 0 && a.wide_out) a.wide_out[r * a.ld_wide_out] = shared;
        }
#pragma unroll
        for (int t = 0; t < CDC_HEAD_MAX_TOWERS; ++t) {
            if (t >= a.n_tower) break;
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < HEAD_TJ; ++j)
                if (lane + 64 * j < a.t[t].K) acc += tx[t][j] * tw[t][j];
            acc = wave_sum(acc);
            if (lane == 0) {
                if (a.t[t].bias) acc += a.t[t].bias[0];
                if (a.wide_x) acc += shared;
                for (int i = 0; i < a.n_addend; ++i) acc += a.addend[i][r * a.ld_addend[i]];
                if (a.sigmoid) acc = 1.f / (1.f + expf(-acc));
                a.out[r * a.ld_out + t] = acc;
            }
        }
        return;
    }
    float shared = 0.f;                                                  // what every tower's logit receives
    if (a.wide_x) {
        shared = head_dot(a.wide_x + r * a.ld_wide, a.wide_w, a.wide_K, lane);
        if (a.wide_bias) shared += a.wide_bias[0];
        if (lane == 0 && a.wide_out) a.wide_out[r * a.ld_wide_out] = shared;
    }
    // `y_logits += other` in the reference's order: the wide term first, then the further addends, each added to the tower's
    // own logit — one fp32 addition per term, like the in-place adds
    for (int t = 0; t < a.n_tower; ++t) {
        const cdc_head_tower& T = a.t[t];
        float acc = head_dot(T.x + r * T.ldx, T.w, T.K, lane);
        if (lane == 0) {
            if (T.bias) acc += T.bias[0];
            if (a.wide_x) acc += shared;
            for (int i = 0; i < a.n_addend; ++i) acc += a.addend[i][r * a.ld_addend[i]];
            if (a.sigmoid) acc = 1.f / (1.f + expf(-acc));
            a.out[r * a.ld_out + t] = acc;
        }
    }
}

extern "C" int cdc_head_fwd(const cdc_head_args* a, void* stream) {
    CDC_CHECK_ARG(a && a->n_tower > 0 && a->n_tower <= CDC_HEAD_MAX_TOWERS && a->M >= 0 && a->out && a->ld_out >= a->n_tower &&
                      a->n_addend >= 0 && a->n_addend <= 2, CDC_E_BADARG, "head_fwd: bad argument");
    for (int t = 0; t < a->n_tower; ++t)
        CDC_CHECK_ARG(a->t[t].x && a->t[t].w && a->t[t].K > 0 && a->t[t].ldx >= a->t[t].K, CDC_E_BADARG, "head_fwd: tower %d malformed", t);
    CDC_CHECK_ARG(!a->wide_x || (a->wide_w && a->wide_K > 0 && a->ld_wide >= a->wide_K), CDC_E_BADARG, "head_fwd: wide term malformed");
    if (a->M == 0) return 0;
    hipLaunchKernelGGL(k_head_fwd, dim3((unsigned)cdc_ceil_div(a->M, HEAD_WAVES)), dim3(HEAD_THREADS), 0, (hipStream_t)stream, *a);
    CDC_LAUNCH_CHECK("head_fwd");
    return 0;
}

// workspace layout per part: [tower 0: K_0 dw | db][tower 1 ...]...[wide: K_w dw | db]; `off` = start of a section
__device__ __forceinline__ int head_section(const cdc_head_args& a, int t) {       // t == n_tower: the wide section
    int off = 0;
    for (int i = 0; i < t; ++i) off += a.t[i].K + 1;
    return off;
}

#define HEAD_BWD_WAVES 8
// NT: the tower loops are unrolled to NT (4 or CDC_HEAD_MAX_TOWERS).  Unrolled to 8 the kernel needs 231 VGPRs (half of them spilled
// scalar registers: eight towers' pointers and strides): two of its waves no longer fit on a SIMD beside the two low-priority waves of
// the background replay slice (2 x 231 + 2 x 64 > 512), its workgroups waited for the slice to retire and the launch took 51 us in the
// C2 step against 19 alone.  At NT = 4: 150.
template <int NT>
__global__ void __launch_bounds__(HEAD_BWD_WAVES * 64) k_head_bwd(const cdc_head_args a, int width) {
    CDC_PRIO_MAIN();
    extern __shared__ float head_sh[];                                   // [HEAD_BWD_WAVES][width]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int part = blockIdx.x;
    const int M = (int)a.M;
    const int per = (M + CDC_ROWDOT_PARTS - 1) / CDC_ROWDOT_PARTS;
    const int r_begin = part * per, r_end = min(r_begin + per, M);
    float* mine = head_sh + wave * width;
    for (int k = lane; k < width; k += 64) mine[k] = 0.f;
    const bool bce = a.bce_y_i16 != nullptr || a.bce_y_f32 != nullptr;
    double loss_part = 0.0;
    const int wide_off = head_section(a, a.n_tower);
    bool small = !a.wide_x || a.wide_K <= 64 * HEAD_KJ;
    for (int t = 0; t < a.n_tower; ++t) small = small && a.t[t].K <= 64 * HEAD_TJ;
    const bool wide_rmw = a.wide_dx && a.accumulate_wide_dx;
    // one row per wave and round (a part holds 16 rows at B = 4096: two rounds of 8 waves): wave w takes rows r_begin + w, + 8, ...; every
    // load of the row — labels, outputs, tower inputs, the wide input, the gradients that are added to — is issued before the
    // first use, so a round is two dependent round trips whatever the number of towers
    for (int r = r_begin + wave; r < r_end; r += HEAD_BWD_WAVES) {
        const int64_t rr = r;
        int64_t c = 0;
        float tgt = 0.f;
        if (bce) {
            c = a.bce_group ? a.bce_group[rr] : 0;
            tgt = a.bce_y_i16 ? (float)a.bce_y_i16[rr] : a.bce_y_f32[rr];
        }
        float o[NT], dout_in[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            o[t] = t < a.n_tower ? a.out[rr * a.ld_out + t] : 0.5f;
            dout_in[t] = (!bce && t < a.n_tower) ? a.d_out[rr * a.ld_dout + t] : 0.f;
        }
        float wx[HEAD_KJ], wold[HEAD_KJ], wk[HEAD_KJ];
        float tx[NT][HEAD_TJ], told[NT][HEAD_TJ], twk[NT][HEAD_TJ];
        if (small) {
#pragma unroll
            for (int j = 0; j < HEAD_KJ; ++j) {
                const int k = lane + 64 * j;
                const bool in = a.wide_x && k < a.wide_K;
                wx[j] = in ? a.wide_x[rr * a.ld_wide + k] : 0.f;
                wold[j] = (in && wide_rmw) ? a.wide_dx[rr * a.ld_wide_dx + k] : 0.f;
                wk[j] = (in && a.wide_dx) ? a.wide_w[k] : 0.f;
            }
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < HEAD_TJ; ++j) {
                    const int k = lane + 64 * j;
                    const bool in = t < a.n_tower && k < a.t[t].K;
                    tx[t][j] = in ? a.t[t].x[rr * a.t[t].ldx + k] : 0.f;
                    told[t][j] = (in && a.t[t].dx && a.t[t].accumulate_dx) ? a.t[t].dx[rr * a.t[t].lddx + k] : 0.f;
                    twk[t][j] = (in && a.t[t].dx) ? a.t[t].w[k] : 0.f;
                }
        }
        // (1) the logit gradient of every tower of this row (every lane forms all of them: n_tower is small)
        if (c < 0 || c >= a.n_tower) c = 0;
        const int own = (int)c;
        float d[NT];
        float dsum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            d[t] = 0.f;
            if (t < a.n_tower) {
                float dout;
                if (bce) {
                    // BCELoss(mean) on the row's own tower and its gradient (cdc_bce_fwd_bwd's arithmetic, operation for operation)
                    if (t == own) {
                        loss_part += (double)((tgt - 1.f) * fmaxf(log1pf(-o[t]), -100.f) - tgt * fmaxf(logf(o[t]), -100.f));
                        dout = a.bce_inv_count * (o[t] - tgt) / fmaxf((1.f - o[t]) * o[t], 1e-12f);
                    } else dout = 0.f;
                } else dout = dout_in[t];
                d[t] = a.sigmoid ? dout * o[t] * (1.f - o[t]) : dout;
                dsum += d[t];                                            // ascending tower order: what the fan-in add formed
            }
        }
        // (2) towers: dx_t = d_t * w_t, dw_t += d_t * x_t, db_t += d_t
        int off = 0;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t >= a.n_tower) break;
            const cdc_head_tower& T = a.t[t];
            const float dt = d[t];
            if (small) {
#pragma unroll
                for (int j = 0; j < HEAD_TJ; ++j) {
                    const int k = lane + 64 * j;
                    if (k >= T.K) break;
                    mine[off + k] += dt * tx[t][j];
                    if (T.dx) {
                        const float v = dt * twk[t][j];
                        T.dx[rr * T.lddx + k] = T.accumulate_dx ? told[t][j] + v : v;
                    }
                }
            } else {
                for (int k = lane; k < T.K; k += 64) {
                    mine[off + k] += dt * T.x[rr * T.ldx + k];
                    if (T.dx) {
                        float* dst = T.dx + rr * T.lddx + k;
                        const float v = dt * T.w[k];
                        *dst = T.accumulate_dx ? *dst + v : v;
                    }
                }
            }
            if (lane == 0) mine[off + T.K] += dt;
            off += T.K + 1;
        }
        // (3) what every tower's logit received gets the sum of the towers' logit gradients
        for (int i = 0; i < a.n_addend; ++i) {
            if (lane == 0 && a.d_addend[i]) {
                float* dst = a.d_addend[i] + rr * a.ld_d_addend[i];
                *dst = a.accumulate_d_addend[i] ? *dst + dsum : dsum;
            }
        }
        if (a.wide_x) {
            if (small) {
#pragma unroll
                for (int j = 0; j < HEAD_KJ; ++j) {
                    const int k = lane + 64 * j;
                    if (k >= a.wide_K) break;
                    mine[wide_off + k] += dsum * wx[j];
                    if (a.wide_dx) {
                        const float v = dsum * wk[j];
                        a.wide_dx[rr * a.ld_wide_dx + k] = wide_rmw ? wold[j] + v : v;
                    }
                }
            } else {
                for (int k = lane; k < a.wide_K; k += 64) {
                    mine[wide_off + k] += dsum * a.wide_x[rr * a.ld_wide + k];
                    if (a.wide_dx) {
                        float* dst = a.wide_dx + rr * a.ld_wide_dx + k;
                        const float v = dsum * a.wide_w[k];
                        *dst = wide_rmw ? *dst + v : v;
                    }
                }
            }
            if (lane == 0) mine[wide_off + a.wide_K] += dsum;
        }
    }
    __shared__ double loss_w[HEAD_BWD_WAVES];
    if (bce && lane == 0) loss_w[wave] = loss_part;                      // every lane of a wave holds the same sum
    __syncthreads();
    if (bce && threadIdx.x == 0) {
        double s_ = 0.0;
        for (int w = 0; w < HEAD_BWD_WAVES; ++w) s_ += loss_w[w];        // wave order = ascending row order inside the part
        a.bce_partial[part] = s_;
    }
    float* ws = a.workspace + (int64_t)part * width;
    for (int k = threadIdx.x; k < width; k += HEAD_BWD_WAVES * 64) {
        float s_ = 0.f;
#pragma unroll
        for (int w = 0; w < HEAD_BWD_WAVES; ++w) s_ += head_sh[w * width + k];
        ws[k] = s_;
    }
}

// one wave per output element: lane l adds parts l, l+64, ... in ascending order, a butterfly adds the 64 lane sums; the last
// block's first wave adds the loss partials
__global__ void __launch_bounds__(HEAD_THREADS) k_head_bwd_final(const cdc_head_args a, int width) {
    CDC_PRIO_MAIN();
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * HEAD_WAVES + (threadIdx.x >> 6);
    if (k == width) {
        if (!a.bce_loss || !(a.bce_y_i16 || a.bce_y_f32)) return;
        double s = 0.0;
        for (int i = lane; i < CDC_ROWDOT_PARTS; i += 64) s += a.bce_partial[i];
        s = wave_sum_d(s);
        if (lane == 0) *a.bce_loss = (float)(s * (double)a.bce_inv_count);
        return;
    }
    if (k > width) return;
    float v[CDC_ROWDOT_PARTS / 64];
#pragma unroll
    for (int i = 0; i < CDC_ROWDOT_PARTS / 64; ++i) v[i] = a.workspace[(int64_t)(lane + 64 * i) * width + k];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CDC_ROWDOT_PARTS / 64; ++i) s += v[i];
    s = wave_sum(s);
    if (lane != 0) return;
    int off = 0;
    for (int t = 0; t < a.n_tower; ++t) {
        const cdc_head_tower& T = a.t[t];
        if (k < off + T.K) { if (T.dw) T.dw[k - off] = s; return; }
        if (k == off + T.K) { if (T.dbias) T.dbias[0] = s; return; }
        off += T.K + 1;
    }
    if (a.wide_x) {
        if (k < off + a.wide_K) { if (a.wide_dw) a.wide_dw[k - off] = s; }
        else if (k == off + a.wide_K) { if (a.wide_dbias) a.wide_dbias[0] = s; }
    }
}

static int head_width(const cdc_head_args* a) {
    int w = 0;
    for (int t = 0; t < a->n_tower; ++t) w += a->t[t].K + 1;
    if (a->wide_x) w += a->wide_K + 1;
    return w;
}
extern "C" int64_t cdc_head_workspace_floats(const cdc_head_args* a) {
    if (!a || a->n_tower <= 0 || a->n_tower > CDC_HEAD_MAX_TOWERS) return -1;
    return (int64_t)CDC_ROWDOT_PARTS * head_width(a);
}

extern "C" int cdc_head_bwd(const cdc_head_args* a, void* stream) {
    CDC_CHECK_ARG(a && a->n_tower > 0 && a->n_tower <= CDC_HEAD_MAX_TOWERS && a->M >= 0 && a->out && a->workspace &&
                      a->n_addend >= 0 && a->n_addend <= 2, CDC_E_BADARG, "head_bwd: bad argument");
    const bool bce = a->bce_y_i16 || a->bce_y_f32;
    CDC_CHECK_ARG(bce || a->d_out, CDC_E_BADARG, "head_bwd: needs the output gradient or the fused loss");
    CDC_CHECK_ARG(!bce || (a->sigmoid && a->bce_loss && a->bce_partial && a->bce_inv_count > 0.f), CDC_E_BADARG,
                  "head_bwd: the fused BCE needs sigmoid outputs, a loss pointer and the partial-sum buffer");
    for (int t = 0; t < a->n_tower; ++t)
        CDC_CHECK_ARG(a->t[t].x && a->t[t].w && a->t[t].K > 0, CDC_E_BADARG, "head_bwd: tower %d malformed", t);
    const int width = head_width(a);
    CDC_CHECK_ARG((size_t)HEAD_BWD_WAVES * width * 4 <= 64 * 1024, CDC_E_TOOBIG, "head_bwd: too many weight-gradient columns for one pass");
    hipStream_t st = (hipStream_t)stream;
    if (a->n_tower <= 4)
        hipLaunchKernelGGL(k_head_bwd<4>, dim3(CDC_ROWDOT_PARTS), dim3(HEAD_BWD_WAVES * 64), HEAD_BWD_WAVES * width * sizeof(float), st, *a, width);
    else
        hipLaunchKernelGGL(k_head_bwd<CDC_HEAD_MAX_TOWERS>, dim3(CDC_ROWDOT_PARTS), dim3(HEAD_BWD_WAVES * 64), HEAD_BWD_WAVES * width * sizeof(float), st, *a, width);
    CDC_LAUNCH_CHECK("head_bwd");
    hipLaunchKernelGGL(k_head_bwd_final, dim3((unsigned)cdc_ceil_div(width + 1, HEAD_WAVES)), dim3(HEAD_THREADS), 0, st, *a, width);
    CDC_LAUNCH_CHECK("head_bwd_final");
    return 0;
}
