// Scalar factors of the SO(3) exponential and logarithm, shared by every retraction / prior / odometry factor of the library.
// A Levenberg-Marquardt or Gauss-Newton step, the error of a pose prior and of an odometry factor are SMALL rotations: the factors
// are evaluated as series in theta^2 (sin^2 theta for the logarithm) there -- a dozen FMAs where the library's sin, cos and acos
// with their argument reduction cost ~1 us of a lone wavefront's chain each (the retraction sits on the serial chain of a
// Gauss-Newton iteration and of every PnP refinement), and without the cancellation of 1 - cos(theta).
#pragma once
#include "tri_math.h"

namespace mqs {

// a = sin(theta) / theta, b = (1 - cos(theta)) / theta^2 from th2 = theta^2.  Series below theta = 0.5 (truncation < 1e-19).
MQS_HD void so3_exp_factors(double th2, double &a, double &b)
{
    if (th2 < 0.25) {
        a = fma(th2, fma(th2, fma(th2, fma(th2, fma(th2, fma(th2, fma(th2, -1.0 / 1307674368000.0, 1.0 / 6227020800.0), -1.0 / 39916800.0),
                1.0 / 362880.0), -1.0 / 5040.0), 1.0 / 120.0), -1.0 / 6.0), 1.0);
        b = fma(th2, fma(th2, fma(th2, fma(th2, fma(th2, fma(th2, fma(th2, -1.0 / 20922789888000.0, 1.0 / 87178291200.0), -1.0 / 479001600.0),
                1.0 / 3628800.0), -1.0 / 40320.0), 1.0 / 720.0), -1.0 / 24.0), 0.5);
    } else {
        const double th = sqrt(th2);
        a = sin(th) / th;
        b = (1.0 - cos(th)) / th2;
    }
}

// k = theta / (2 sin theta) for the logarithm w = k vee(R - R^T), from c = cos theta = (trace R - 1) / 2 and
// s2 = sin^2 theta = |vee(R - R^T)|^2 / 4.  Below 0.1 rad: asin(s) / (2 s) as a series in s2 (seven terms: 1.4e-16 relative at the
// switch) -- more accurate there than acos of a cosine next to 1.
MQS_HD double so3_log_factor(double c, double s2)
{
    c = fmin(1.0, fmax(-1.0, c));
    if (c > 0.0 && s2 < 0.01) {
        double p = 143.0 / 10240.0;
        p = fma(p, s2, 231.0 / 13312.0);
        p = fma(p, s2, 63.0 / 2816.0);
        p = fma(p, s2, 35.0 / 1152.0);
        p = fma(p, s2, 5.0 / 112.0);
        p = fma(p, s2, 3.0 / 40.0);
        p = fma(p, s2, 1.0 / 6.0);
        return 0.5 * fma(p, s2, 1.0);
    }
    const double th = acos(c);
    return (th < 1e-10) ? 0.5 : th / (2.0 * sin(th));
}

}  // namespace mqs
