/*
 * paris_oracle.h -- CPU restatement of the hzdr/PARIS OpenMP hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it; the product (paris_amd/, include/paris_hip.h) never links, imports or calls it.
 *
 * Every function cites the reference file:line it restates (paths relative to /root/reference).
 * Arithmetic is fp32 with one rounding per operation, in the reference's operation order; build
 * with -ffp-contract=off and no -march so no FMA contraction happens (the reference's CMake sets
 * neither, src/CMakeLists.txt:60-98, CMakeLists.txt:55-90).
 *
 * Differences from the reference that do not change results:
 *   - 64-bit voxel indexing (reference: u32);
 *   - no function-local statics: constants are passed per call;
 *   - the row FFT is this file's own fp32 radix-2 transform (the reference calls FFTW3f, a
 *     third-party library that is absent from /root/reference and from this image), so filter
 *     results agree with the reference only to FFT rounding, as they do between FFTW builds.
 */
#ifndef PARIS_ORACLE_H_
#define PARIS_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/geometry.h:30-46 */
typedef struct {
    uint32_t n_row, n_col;
    float l_px_row, l_px_col;
    float delta_s, delta_t;
    float d_so, d_od;
    float delta_phi;
} po_detector_geometry;

/* src/geometry.h:48-57 */
typedef struct {
    uint32_t dim_x, dim_y, dim_z;
    float l_vx_x, l_vx_y, l_vx_z;
} po_volume_geometry;

/* src/region_of_interest.h:30-38 */
typedef struct {
    uint32_t x1, x2, y1, y2, z1, z2;
} po_region_of_interest;

/* src/geometry.cpp:36-84 */
void po_calculate_volume_geometry(const po_detector_geometry* det, po_volume_geometry* out);
/* src/geometry.cpp:86-130 */
void po_apply_roi(const po_volume_geometry* vol, const po_region_of_interest* roi, po_volume_geometry* out);

/* src/weighting.cpp:32-45 (wrapper constants) */
void po_weight_constants(const po_detector_geometry* det, float* h_min, float* v_min, float* d_sd);
/* src/openmp/weighting.cpp:32-57 */
void po_weight(float* p, uint32_t dim_x, uint32_t dim_y, float h_min, float v_min, float d_sd,
               float l_px_row, float l_px_col);

/* src/filtering.cpp:37 */
uint32_t po_filter_size(uint32_t n_row);
/* src/openmp/filtering.cpp:52-73 */
void po_make_filter_real(float* r, uint32_t size, float tau);
/* src/openmp/filtering.cpp:139-165; k receives size/2+1 real values (the reference stores each in re and im) */
void po_make_filter(float* k, uint32_t size, float tau);
/* src/openmp/filtering.cpp:155-162 applied to a spectrum supplied by the caller (re/im interleaved,
 * size/2+1 bins): lets a test plug in another FFT for the one-off transform */
void po_make_filter_from_spectrum(const float* spec, float* k, uint32_t size, float tau);
/* src/openmp/filtering.cpp:167-219; p is dim_x x n_col row-major, filtered in place */
void po_apply_filter(float* p, uint32_t dim_x, uint32_t n_col, const float* k, uint32_t filter_size);

/* src/backprojection.cpp:37-69 (wrapper: angle -> sin/cos, offsets in mm) */
void po_backproject_constants(const po_detector_geometry* det, uint32_t idx, int enable_angles, float phi_in,
                              float* sin_out, float* cos_out, float* delta_s_mm, float* delta_t_mm);
/* src/openmp/backprojection.cpp:86-199 */
void po_backproject(float* vol, uint32_t v_dim_x, uint32_t v_dim_y, uint32_t v_dim_z,
                    const float* p, uint32_t p_dim_x, uint32_t p_dim_y, uint32_t v_offset,
                    const po_detector_geometry* det, const po_volume_geometry* vol_geo,
                    int enable_roi, const po_region_of_interest* roi,
                    float sin_phi, float cos_phi, float delta_s_mm, float delta_t_mm);

/* helpers for the known-answer tests */
uint64_t po_fnv1a64(const void* data, size_t n);
void po_lcg_fill(float* p, size_t n, uint32_t idx);
int po_num_threads(void);
void po_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
