// libblsw.so, one translation unit per kernel family (see kcommon.hpp, build.py).
#include "kcommon.hpp"
#include "values.hpp"
#include "vsign.hpp"

namespace blsw {

// native signer (bls.rs:411-425, 183-195): lanes [0, n) sig_i = sk_i * H(msg_i) (H projective in ws.h), lanes [n, 2n)
// pk_i = sk_i * g1. Outputs (each optional): compressed bytes and affine Montgomery limbs; status[i] (SIGN_*)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_sign(uint64_t n, Workspace ws, const uint8_t* __restrict__ sk32, uint8_t* sig96, uint64_t* sig_xy, uint8_t* pk48,
                                             uint64_t* pk_xy, int32_t* status) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * n) return;
    const uint64_t i = t < n ? t : t - n;
    uint32_t k[8];
    int st = sk_from_le32(sk32 + i * 32, k);
    if (t < n) {
        Fp2 x = fp2_zero(), y = fp2_zero();
        bool inf = true;
        if (st == SIGN_OK) {
            Proj<OpsFp2> h = ld_proj2(ws.h + i, n);
            if (!fp2_is_zero(h.z)) {  // sig = sk * H(m): joint ladder over the four base-|x| digits of sk (vsign.hpp)
                // homogeneous (x, y, z) = affine (x / z, y / z) = Jacobian (x z, y z^2, z)
                const Jac2 q = {fp2_mul_inl(h.x, h.z), fp2_mul_inl(h.y, v_sqr(h.z)), h.z};
                const Jac2 acc = v_g2_mul_gls(ParkRows{ws.coeff_h + i, n}, q, k);
                if (!fp2_is_zero(acc.z)) {
                    const Fp2 ai = fp2_inv_inl(acc.z), ai2 = v_sqr(ai);
                    x = fp2_mul_inl(acc.x, ai2);
                    y = fp2_mul_inl(acc.y, fp2_mul_inl(ai2, ai));
                    inf = false;
                }
            }
        }
        if (sig_xy) {
            Fp* o = reinterpret_cast<Fp*>(sig_xy + i * 24);
            st_fp(o, x.c0);
            st_fp(o + 1, x.c1);
            st_fp(o + 2, y.c0);
            st_fp(o + 3, y.c1);
        }
        if (sig96) g2_encode(x, y, inf, sig96 + i * 96);
        status[i] = st;
    } else {
        Fp x = fp_zero(), y = fp_zero();
        bool inf = true;
        if (st == SIGN_OK) {  // pk = sk * g1: fixed-base windows (vsign.hpp)
            const Jac1v acc = v1_mul_g1_fixed(k);
            if (!fp_is_zero(acc.z)) {
                const Fp ai = fp_inv(acc.z), ai2 = fp_sqr(ai);
                x = fp_mul(acc.x, ai2);
                y = fp_mul(acc.y, fp_mul(ai2, ai));
                inf = false;
            }
        }
        if (pk_xy) {
            Fp* o = reinterpret_cast<Fp*>(pk_xy + i * 12);
            st_fp(o, x);
            st_fp(o + 1, y);
        }
        if (pk48) g1_encode(x, y, inf, pk48 + i * 48);
    }
}

}  // namespace blsw
