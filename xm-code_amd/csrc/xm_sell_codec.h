// xm_sell_codec.h — the view-graph block codec shared by the sliced-ELL layouts (xm_sell.hip, xm_sell2.hip): an off-diagonal block
// -w * M (M a rotation) is stored as the quaternion of M scaled by sqrt(2 w) and rebuilt in registers (rationale: xm_sell.h).
#pragma once

#include <cmath>

#include "xm_common.h"

namespace xm {

// block -w * M (M a rotation, row-major) -> quaternion of M scaled by sqrt(2 w): (a; b, c, d) with a the scalar part.  The product
// kernel rebuilds the block as -R(q) from products of pairs (qw_sell_body).  Shepperd's branch selection keeps the divisor >= 1.
__host__ __device__ inline void block_to_quat(const double (&b)[9], double (&q)[4]) {
    double ss = 0.0;
    for (int e = 0; e < 9; ++e) ss += b[e] * b[e];
    const double w = sqrt(ss / 3.0);
    if (!(w > 0.0)) { q[0] = q[1] = q[2] = q[3] = 0.0; return; }
    double M[9];
    for (int e = 0; e < 9; ++e) M[e] = -b[e] / w;
    const double tr = M[0] + M[4] + M[8];
    double a, x, y, z;
    if (tr > 0.0) {
        const double S = sqrt(tr + 1.0) * 2.0;
        a = 0.25 * S; x = (M[7] - M[5]) / S; y = (M[2] - M[6]) / S; z = (M[3] - M[1]) / S;
    } else if (M[0] > M[4] && M[0] > M[8]) {
        const double S = sqrt(1.0 + M[0] - M[4] - M[8]) * 2.0;
        a = (M[7] - M[5]) / S; x = 0.25 * S; y = (M[1] + M[3]) / S; z = (M[2] + M[6]) / S;
    } else if (M[4] > M[8]) {
        const double S = sqrt(1.0 + M[4] - M[0] - M[8]) * 2.0;
        a = (M[2] - M[6]) / S; x = (M[1] + M[3]) / S; y = 0.25 * S; z = (M[5] + M[7]) / S;
    } else {
        const double S = sqrt(1.0 + M[8] - M[0] - M[4]) * 2.0;
        a = (M[3] - M[1]) / S; x = (M[2] + M[6]) / S; y = (M[5] + M[7]) / S; z = 0.25 * S;
    }
    const double nrm = sqrt((a * a + x * x) + (y * y + z * z));
    const double sc = sqrt(2.0 * w) / nrm;
    q[0] = a * sc; q[1] = x * sc; q[2] = y * sc; q[3] = z * sc;
}
// the block the product kernel works with, rebuilt from the stored quaternion (same expression tree as qw_sell_body)
__host__ __device__ inline void quat_to_block(double a, double b, double c, double d, double (&q)[9]) {
    const double aa = a * a, bb = b * b, cc = c * c, dd = d * d;
    const double n2 = 0.5 * ((aa + bb) + (cc + dd));
    const double ad = a * d, ac = a * c, ab = a * b;
    q[0] = n2 - (aa + bb); q[4] = n2 - (aa + cc); q[8] = n2 - (aa + dd);
    q[1] = fma(-b, c, ad);  q[3] = -fma(b, c, ad);
    q[2] = -fma(b, d, ac);  q[6] = fma(-b, d, ac);
    q[5] = fma(-c, d, ab);  q[7] = -fma(c, d, ab);
}


}  // namespace xm
