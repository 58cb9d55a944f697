// libblsw.so, one translation unit per kernel family (see kcommon.hpp, build.py).
#include "kcommon.hpp"
#include "values.hpp"

namespace blsw {

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_map_values(Group g) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * g.N) return;
    const uint32_t which = t >= g.N;
    const uint64_t I = which ? t - g.N : t, N = g.N;
    const Proj<OpsFp2> q = v_map_to_curve(ld_fp2(g.ws.u + (uint64_t)(2 * which) * N + I, N));
    Fp* o = g.ws.q + (uint64_t)(6 * which) * N + I;
    st_fp(o, q.x.c0);
    st_fp(o + N, q.x.c1);
    st_fp(o + 2 * N, q.y.c0);
    st_fp(o + 3 * N, q.y.c1);
    st_fp(o + 4 * N, q.z.c0);
    st_fp(o + 5 * N, q.z.c1);
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_cofactor_values(Group g) {
    uint64_t I = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= g.N) return;
    const uint64_t N = g.N;
    // Q0, Q1 leave the isogeny as (x, y, 1) or (0, 0, 0) (hasher.rs:339-345): affine points or the identity
    Jac2 r;
    {
        const Proj<OpsFp2> q0 = ld_proj2(g.ws.q + I, N);
        r = {q0.x, q0.y, q0.z};
        if (fp2_is_zero(q0.z)) r = {fp2_one(), fp2_one(), fp2_zero()};
    }
    {
        const Proj<OpsFp2> q1 = ld_proj2(g.ws.q + 6 * N + I, N);
        if (!fp2_is_zero(q1.z)) r = v_add_mixed(r, q1.x, q1.y);  // hasher.rs:656 (handles Q0 = +-Q1 and Q0 = 0)
    }
    Proj<OpsFp2> h = {fp2_zero(), fp2_one(), fp2_zero()};
    if (!fp2_is_zero(r.z)) {
        const Jac2 acc = v_clear_cofactor(ParkRows{g.ws.coeff_h + I, N}, r);  // hasher.rs:664-673 (same point as h_eff * r)
        if (!fp2_is_zero(acc.z)) {  // (X / Z^2, Y / Z^3) as homogeneous (X Z, Y, Z^3)
            h.x = fp2_mul_inl(acc.x, acc.z);
            h.y = acc.y;
            h.z = fp2_mul_inl(v_sqr(acc.z), acc.z);
        }
    }
    Fp* o = g.ws.h + I;
    st_fp(o, h.x.c0);
    st_fp(o + N, h.x.c1);
    st_fp(o + 2 * N, h.y.c0);
    st_fp(o + 3 * N, h.y.c1);
    st_fp(o + 4 * N, h.z.c0);
    st_fp(o + 5 * N, h.z.c1);
}

// input decode: lanes [0, n) decompress pk (48 B), lanes [n, 2n) decompress sig (96 B); status[i][0] / status[i][1]
__global__ __launch_bounds__(64) void k_decode(const uint8_t* __restrict__ pk48, const uint8_t* __restrict__ sig96, uint64_t n, uint64_t* pk_xy,
                                               uint64_t* sig_xy, int32_t* status) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * n) return;
    if (t < n) {
        Fp x, y;
        int st = g1_decode(pk48 + t * 48, x, y);
        Fp* o = reinterpret_cast<Fp*>(pk_xy + t * 12);
        st_fp(o, x);
        st_fp(o + 1, y);
        status[2 * t] = st;
    } else {
        uint64_t i = t - n;
        Fp2 x, y;
        int st = g2_decode(sig96 + i * 96, x, y);
        Fp* o = reinterpret_cast<Fp*>(sig_xy + i * 24);
        st_fp(o, x.c0);
        st_fp(o + 1, x.c1);
        st_fp(o + 2, y.c0);
        st_fp(o + 3, y.c1);
        status[2 * i + 1] = st;
    }
}

// Signature::aggregate / PublicKey::aggregate (bls.rs:288-300, 183-195). Phase 1: one lane per compressed point (try_from: flags, x < p, on
// curve, subgroup), affine coordinates and status into the workspace. Phase 2: one lane per list adds its k points and serialises the sum.
__global__ __launch_bounds__(64) void k_decode_points(uint32_t group, const uint8_t* __restrict__ in, uint64_t m, Fp* xy, int32_t* st) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    if (group == 1) {
        Fp x, y;
        st[t] = g1_decode(in + t * 48, x, y);
        st_fp(xy + 2 * t, x);
        st_fp(xy + 2 * t + 1, y);
    } else {
        Fp2 x, y;
        st[t] = g2_decode(in + t * 96, x, y);
        st_fp(xy + 4 * t, x.c0);
        st_fp(xy + 4 * t + 1, x.c1);
        st_fp(xy + 4 * t + 2, y.c0);
        st_fp(xy + 4 * t + 3, y.c1);
    }
}
__global__ __launch_bounds__(64) void k_sum_points(uint32_t group, const Fp* __restrict__ xy, const int32_t* __restrict__ pst, uint32_t k, uint64_t n, uint8_t* out,
                                                   int32_t* status) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int bad = DEC_OK;
    for (uint32_t j = 0; j < k && bad == DEC_OK; j++) {
        const int s = pst[i * k + j];
        if (s != DEC_OK && s != DEC_IDENTITY) bad = s;
    }
    status[i] = bad;
    const uint32_t bytes = group == 1 ? 48u : 96u;
    uint8_t* o = out + i * bytes;
    if (bad != DEC_OK) {
        for (uint32_t b = 0; b < bytes; b++) o[b] = 0;
        return;
    }
    if (group == 1) {
        Jac1v acc = {fp_one(), fp_one(), fp_zero()};
#pragma unroll 1
        for (uint32_t j = 0; j < k; j++) {
            if (pst[i * k + j] == DEC_IDENTITY) continue;
            const Fp* p = xy + 2 * (i * k + j);
            acc = jac1v_add_mixed(acc, ld_fp(p), ld_fp(p + 1));
        }
        const bool inf = fp_is_zero(acc.z);
        const Fp zi = fp_inv(acc.z), zi2 = fp_sqr(zi);
        g1_encode(fp_mul(acc.x, zi2), fp_mul(acc.y, fp_mul(zi2, zi)), inf, o);
    } else {
        Jac2 acc = {fp2_one(), fp2_one(), fp2_zero()};
#pragma unroll 1
        for (uint32_t j = 0; j < k; j++) {
            if (pst[i * k + j] == DEC_IDENTITY) continue;
            const Fp* p = xy + 4 * (i * k + j);
            acc = jac2_add_mixed(acc, {ld_fp(p), ld_fp(p + 1)}, {ld_fp(p + 2), ld_fp(p + 3)});
        }
        const bool inf = fp2_is_zero(acc.z);
        const Fp2 zi = fp2_inv(acc.z), zi2 = fp2_sqr(zi);
        g2_encode(fp2_mul(acc.x, zi2), fp2_mul(acc.y, fp2_mul(zi2, zi)), inf, o);
    }
}

// H(m) projective -> affine (hash_to_g2 batch output)
__global__ __launch_bounds__(64) void k_h_to_affine(uint64_t n, Workspace ws, uint64_t* d_out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Proj<OpsFp2> h = ld_proj2(ws.h + i, n);
    Fp2 zi = fp2_inv(h.z);
    Fp2 x = fp2_mul(h.x, zi), y = fp2_mul(h.y, zi);
    Fp* o = reinterpret_cast<Fp*>(d_out + i * 24);
    st_fp(o, x.c0);
    st_fp(o + 1, x.c1);
    st_fp(o + 2, y.c0);
    st_fp(o + 3, y.c1);
}

}  // namespace blsw
