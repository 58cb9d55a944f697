// Device side of the value-only entries: vcurve.hpp plus the kernels' common definitions.
#pragma once
#include "kcommon.hpp"
#include "vcurve.hpp"

namespace blsw {

// Jacobian points of a lane parked in workspace rows (element-major: row * N + lane, coalesced): the line-coefficient area of
// prepare_g2(H(m)), which the value-only entries do not use
struct ParkRows {
    Fp* p;  // first row, this lane
    uint64_t n;
    __device__ __forceinline__ void st(int slot, const Jac2& v) const {
        Fp* o = p + (uint64_t)(6 * slot) * n;
        st_fp(o, v.x.c0);
        st_fp(o + n, v.x.c1);
        st_fp(o + 2 * n, v.y.c0);
        st_fp(o + 3 * n, v.y.c1);
        st_fp(o + 4 * n, v.z.c0);
        st_fp(o + 5 * n, v.z.c1);
    }
    __device__ __forceinline__ Jac2 ld(int slot) const {
        const Fp* o = p + (uint64_t)(6 * slot) * n;
        return {ld_fp2(o, n), ld_fp2(o + 2 * n, n), ld_fp2(o + 4 * n, n)};
    }
};

}  // namespace blsw
