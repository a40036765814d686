// Host check of sceneego_amd/csrc/fft24.h (the in-register 24-point transform of conv3d_fft7.hip) against a naive float64 DFT.
// Built and run by tests/test_host_logic.py::test_fft24_header_matches_naive_dft with g++; prints "max_err <value>" per direction.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "../../sceneego_amd/csrc/fft24.h"

int main() {
    double worst = 0.0;
    for (int dir = 0; dir < 2; ++dir) {
        double max_err = 0.0;
        for (int trial = 0; trial < 50; ++trial) {
            float re[24], im[24];
            double xr[24], xi[24];
            for (int n = 0; n < 24; ++n) {
                xr[n] = re[n] = (float)(rand() / (double)RAND_MAX - 0.5);
                xi[n] = im[n] = (float)(rand() / (double)RAND_MAX - 0.5);
            }
            if (dir) se_fft24<true>(re, im); else se_fft24<false>(re, im);
            for (int k = 0; k < 24; ++k) {
                double sr = 0, si = 0;
                for (int n = 0; n < 24; ++n) {
                    const double a = (dir ? 2.0 : -2.0) * M_PI * ((n * k) % 24) / 24.0;
                    sr += xr[n] * cos(a) - xi[n] * sin(a);
                    si += xr[n] * sin(a) + xi[n] * cos(a);
                }
                max_err = fmax(max_err, fmax(fabs(sr - re[k]), fabs(si - im[k])));
            }
        }
        printf("max_err %s %.3e\n", dir ? "inverse" : "forward", max_err);
        worst = fmax(worst, max_err);
    }
    return worst < 5e-6 ? 0 : 1;
}
