/* oracle/orlg_oracle_osnr.h -- TEST INFRASTRUCTURE: flattened inputs of the GN-model GSNR routine
 * (examples/calculate_osnr.py:9-56).  Same layout as include/orlg.h orlg_osnr_batch. */
#ifndef ORLG_ORACLE_OSNR_H
#define ORLG_ORACLE_OSNR_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct orc_osnr_batch {
    int32_t num_checks, num_links, num_spans, num_services;
    const int32_t *check_link_off;        /* [num_checks+1] links of each check (= current_service.path.links) */
    const int32_t *link_span_off;         /* [num_links+1]  spans of each link */
    const int32_t *link_svc_off;          /* [num_links+1]  running_services of each link, list order */
    const double *bandwidth, *center_frequency, *launch_power; /* [num_checks] current service: Hz, Hz, W */
    const double *span_length_km, *span_attenuation, *span_noise_figure; /* [num_spans] km, 1/m, linear */
    const double *svc_bandwidth, *svc_center_frequency;        /* [num_services] */
    const int32_t *svc_se;                /* [num_services] current_modulation.spectral_efficiency 1..6 */
    const uint8_t *svc_is_self;           /* [num_services] entry is the current service itself */
} orc_osnr_batch;
void orc_gn_osnr(const orc_osnr_batch *b, double *gsnr_db);
#ifdef __cplusplus
}
#endif
#endif
