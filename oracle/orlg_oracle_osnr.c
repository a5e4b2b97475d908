/*
 * oracle/orlg_oracle_osnr.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see orlg_oracle.h).
 *
 * CPU restatement of the closed-form GN-model GSNR routine examples/calculate_osnr.py:9-56 (calculate_osnr), line by
 * line, INCLUDING its quirk: `sum_phi += phi` also runs for the list entry that IS the current service and then adds
 * the stale `phi` of the previously visited interferer (calculate_osnr.py:31-46); `phi` is initialised once per call
 * (:16) and survives across spans and links.
 *
 * PARITY UNPINNED BY THE REFERENCE: the routine has no caller and no test there, and cannot even be imported
 * (SURVEY.md 0.3, 8c).  tests/golden/osnr_grid.npz was produced by executing the function's own text with its broken
 * import lines dropped (tests/golden/make_golden.py gen_osnr); span.attenuation_normalized / noise_figure_normalized are
 * taken as plain inputs (1/m and linear).
 */
#include <math.h>
#include <stdint.h>

#include "orlg_oracle_osnr.h"

void orc_gn_osnr(const orc_osnr_batch *b, double *gsnr_db) {
    const double beta_2 = -21.3e-27, gamma = 1.3e-3, h_plank = 6.626e-34;
    const double pi = 3.141592653589793;
    const double phi_modulation_format[6] = {1, 1, 2.0 / 3, 17.0 / 25, 69.0 / 100, 13.0 / 21};
    for (int m = 0; m < b->num_checks; m++) {
        const double bw = b->bandwidth[m], fc = b->center_frequency[m], pw = b->launch_power[m];
        double acc_gsnr = 0, l_eff_a = 0, l_eff = 0, phi = 0, sum_phi = 0, power_ase = 0, power_nli_span = 0;
        for (int l = b->check_link_off[m]; l < b->check_link_off[m + 1]; l++) {
            for (int s = b->link_span_off[l]; s < b->link_span_off[l + 1]; s++) {
                const double att = b->span_attenuation[s], len = b->span_length_km[s], nf = b->span_noise_figure[s];
                l_eff_a = 1 / (2 * att);
                l_eff = (1 - exp(-2 * att * len * 1e3)) / (2 * att);
                sum_phi = asinh(pi * pi * fabs(beta_2) * (bw * bw) / (4 * att));
                for (int i = b->link_svc_off[l]; i < b->link_svc_off[l + 1]; i++) {
                    if (!b->svc_is_self[i]) {
                        const double sb = b->svc_bandwidth[i], sf = b->svc_center_frequency[i];
                        phi = (asinh(pi * pi * fabs(beta_2) * l_eff_a * sb * (sf - fc + (sb / 2))) -
                               asinh(pi * pi * fabs(beta_2) * l_eff_a * sb * (sf - fc - (sb / 2)))) -
                              (phi_modulation_format[b->svc_se[i] - 1] * (sb / fabs(sf - fc)) * 5 / 3 * (l_eff / (len * 1e3)));
                    }
                    sum_phi += phi;
                }
                power_nli_span = pow(pw / bw, 3) * (8 / (27 * pi * fabs(beta_2))) * (gamma * gamma) * l_eff * sum_phi * bw;
                power_ase = bw * h_plank * fc * (exp(2 * att * len * 1e3) - 1) * nf;
                acc_gsnr += 1 / (pw / (power_ase + power_nli_span));
            }
        }
        gsnr_db[m] = 10 * log10(1 / acc_gsnr);
    }
}
