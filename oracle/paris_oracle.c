/*
 * paris_oracle.c -- CPU restatement of the hzdr/PARIS OpenMP hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see paris_oracle.h). Parity pin: the known-answer values of
 * SURVEY.md section 8c (tests/test_oracle_kat.py); the reference itself cannot be built in this image
 * (it needs fftw3.h, Boost.Log and GLADOS, none of which is installed), see DESIGN.md.
 * PARITY UNPINNED in the strict sense: the reference holds no tests, fixtures or golden vectors for this path, and the
 * section-8c values were recorded from a survey-session build of the reference's OpenMP sources against stand-in
 * fftw3.h / Boost.Log headers with MKL's FFTW3 interface -- not from anything the reference ships or that can be re-run here.
 *
 * Citations are reference file:line, relative to /root/reference.
 */
#define _GNU_SOURCE
#include "paris_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------
 * geometry -- src/geometry.cpp
 * ---------------------------------------------------------------------------------------------- */

/* src/geometry.cpp:36-67 (make_volume_geometry), :71-84 */
void po_calculate_volume_geometry(const po_detector_geometry* det, po_volume_geometry* out)
{
    const float n_row = (float)det->n_row;
    const float l_px_row = det->l_px_row;
    const float delta_s = fabsf(det->delta_s * l_px_row); /* :44, offset is given in pixels */

    const float n_col = (float)det->n_col;
    const float l_px_col = det->l_px_col;
    const float delta_t = fabsf(det->delta_t * l_px_col); /* :48 */

    const float d_so = fabsf(det->d_so);       /* :50 */
    const float d_sd = fabsf(det->d_od) + d_so; /* :51 */

    const float alpha = atanf((((n_row * l_px_row) / 2.f) + delta_s) / d_sd); /* :54 */
    const float r = d_so * sinf(alpha);                                        /* :55 */

    out->l_vx_x = r / ((((n_row * l_px_row) / 2.f) + delta_s) / l_px_row); /* :57 */
    out->l_vx_y = out->l_vx_x;

    out->dim_x = (uint32_t)((2.f * r) / out->l_vx_x); /* :60 */
    out->dim_y = out->dim_x;

    out->l_vx_z = out->l_vx_x; /* :64 */
    out->dim_z = (uint32_t)(((n_col * l_px_col / 2.f) + delta_t) * (d_so / d_sd) * (2.f / out->l_vx_z)); /* :65 */
}

/* src/geometry.cpp:86-130; on invalid ROI the input geometry is returned unchanged */
void po_apply_roi(const po_volume_geometry* vol, const po_region_of_interest* roi, po_volume_geometry* out)
{
    *out = *vol;
    if(roi->x1 < roi->x2 && roi->y1 < roi->y2 && roi->z1 < roi->z2) /* :96 */
    {
        uint32_t dim_x = roi->x2 - roi->x1;
        uint32_t dim_y = roi->y2 - roi->y1;
        uint32_t dim_z = roi->z2 - roi->z1;
        if(roi->x1 == 0) ++dim_x; /* :102-107 */
        if(roi->y1 == 0) ++dim_y;
        if(roi->z1 == 0) ++dim_z;
        if(dim_x <= vol->dim_x && dim_y <= vol->dim_y && dim_z <= vol->dim_z) /* :109 */
        {
            out->dim_x = dim_x;
            out->dim_y = dim_y;
            out->dim_z = dim_z;
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * weighting -- src/weighting.cpp, src/openmp/weighting.cpp
 * ---------------------------------------------------------------------------------------------- */

/* src/weighting.cpp:37-42 */
void po_weight_constants(const po_detector_geometry* det, float* h_min, float* v_min, float* d_sd)
{
    const float n_row_f = (float)det->n_row;
    const float n_col_f = (float)det->n_col;
    *h_min = (det->delta_s * det->l_px_row) - ((n_row_f * det->l_px_row) / 2);
    *v_min = (det->delta_t * det->l_px_col) - ((n_col_f * det->l_px_col) / 2);
    *d_sd = fabsf(det->d_so) + fabsf(det->d_od);
}

/* src/openmp/weighting.cpp:36-55 */
void po_weight(float* p, uint32_t dim_x, uint32_t dim_y, float h_min, float v_min, float d_sd,
               float l_px_row, float l_px_col)
{
    #pragma omp parallel for
    for(uint32_t t = 0; t < dim_y; ++t)
    {
        for(uint32_t s = 0; s < dim_x; ++s)
        {
            const size_t coord = (size_t)s + (size_t)t * dim_x;
            const float s_f = (float)s;
            const float t_f = (float)t;
            const float h_s = (l_px_row / 2) + s_f * l_px_row + h_min; /* :48 */
            const float v_t = (l_px_col / 2) + t_f * l_px_col + v_min; /* :49 */
            const float w_st = d_sd / sqrtf(d_sd * d_sd + h_s * h_s + v_t * v_t); /* :52 */
            p[coord] *= w_st;
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * filtering -- src/filtering.cpp, src/openmp/filtering.cpp
 * ---------------------------------------------------------------------------------------------- */

/* src/filtering.cpp:37: 2 * 2^ceil(log2(n_row)) */
uint32_t po_filter_size(uint32_t n_row)
{
    return (uint32_t)(2 * pow(2.0, ceil(log2((double)n_row))));
}

/* src/openmp/filtering.cpp:52-73 */
void po_make_filter_real(float* r, uint32_t size, float tau)
{
    const int32_t j0 = -((int32_t)size - 2) / 2; /* :55 */
    const float pi_f = (float)M_PI;
    for(uint32_t x = 0; x < size; ++x)
    {
        const int32_t j = j0 + (int32_t)x;
        if(j == 0)
            r[x] = (1.f / 8.f) * (1.f / powf(tau, 2.f)); /* :64 */
        else if(j % 2 == 0)
            r[x] = 0.f; /* :68 */
        else
            r[x] = -(1.f / (2.f * (float)(j * j) * (pi_f * pi_f) * (tau * tau))); /* :70 */
    }
}

/* In-place iterative radix-2 complex FFT, fp32 data, twiddles rounded from double.
 * Stands in for FFTW3f (src/openmp/filtering.cpp:149,199,201-204), which is not in /root/reference.
 * sign = -1: forward (FFTW_FORWARD, r2c convention), +1: backward (unnormalised, c2r convention). */
static void fft_c2c(float* re, float* im, uint32_t n, int sign, const float* tw_re, const float* tw_im)
{
    /* bit reversal */
    for(uint32_t i = 1, j = 0; i < n; ++i)
    {
        uint32_t bit = n >> 1;
        for(; j & bit; bit >>= 1)
            j ^= bit;
        j ^= bit;
        if(i < j)
        {
            float t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for(uint32_t len = 2; len <= n; len <<= 1)
    {
        const uint32_t half = len >> 1;
        const uint32_t step = n / len;
        for(uint32_t i = 0; i < n; i += len)
        {
            for(uint32_t k = 0; k < half; ++k)
            {
                const float wr = tw_re[k * step];
                const float wi = sign < 0 ? tw_im[k * step] : -tw_im[k * step];
                const float xr = re[i + k + half];
                const float xi = im[i + k + half];
                const float vr = xr * wr - xi * wi;
                const float vi = xr * wi + xi * wr;
                const float ur = re[i + k];
                const float ui = im[i + k];
                re[i + k] = ur + vr;
                im[i + k] = ui + vi;
                re[i + k + half] = ur - vr;
                im[i + k + half] = ui - vi;
            }
        }
    }
}

/* twiddles exp(-2 pi i k / n), k < n/2 */
static void make_twiddles(float* tw_re, float* tw_im, uint32_t n)
{
    for(uint32_t k = 0; k < n / 2; ++k)
    {
        const double a = -2.0 * M_PI * (double)k / (double)n;
        tw_re[k] = (float)cos(a);
        tw_im[k] = (float)sin(a);
    }
}

/* src/openmp/filtering.cpp:155-162 */
void po_make_filter_from_spectrum(const float* spec, float* k, uint32_t size, float tau)
{
    const uint32_t size_trans = size / 2 + 1;
    for(uint32_t x = 0; x < size_trans; ++x)
    {
        const float k0 = spec[2 * x];
        const float k1 = spec[2 * x + 1];
        k[x] = tau * fabsf(sqrtf(powf(k0, 2.f) + powf(k1, 2.f))); /* :157 */
    }
}

/* src/openmp/filtering.cpp:139-165 */
void po_make_filter(float* k, uint32_t size, float tau)
{
    const uint32_t size_trans = size / 2 + 1;
    float* re = (float*)calloc(size, sizeof(float));
    float* im = (float*)calloc(size, sizeof(float));
    float* tw = (float*)malloc(sizeof(float) * size);
    float* spec = (float*)malloc(sizeof(float) * 2 * size_trans);

    po_make_filter_real(re, size, tau); /* :151 */
    make_twiddles(tw, tw + size / 2, size);
    fft_c2c(re, im, size, -1, tw, tw + size / 2); /* :153 */
    for(uint32_t x = 0; x < size_trans; ++x)
    {
        spec[2 * x] = re[x];
        spec[2 * x + 1] = im[x];
    }
    po_make_filter_from_spectrum(spec, k, size, tau);

    free(re); free(im); free(tw); free(spec);
}

/* src/openmp/filtering.cpp:167-219: expand (:75-90), r2c, do_filtering (:92-105), c2r, shrink (:107-118),
 * normalize (:120-131) */
void po_apply_filter(float* p, uint32_t dim_x, uint32_t n_col, const float* k, uint32_t filter_size)
{
    const uint32_t n = filter_size;
    const uint32_t size_trans = n / 2 + 1;
    float* tw = (float*)malloc(sizeof(float) * n);
    make_twiddles(tw, tw + n / 2, n);

    #pragma omp parallel
    {
        float* re = (float*)malloc(sizeof(float) * n);
        float* im = (float*)malloc(sizeof(float) * n);
        #pragma omp for
        for(uint32_t y = 0; y < n_col; ++y)
        {
            float* row = p + (size_t)y * dim_x;
            /* expand :84-87 */
            for(uint32_t x = 0; x < n; ++x)
            {
                re[x] = x < dim_x ? row[x] : 0.f;
                im[x] = 0.f;
            }
            fft_c2c(re, im, n, -1, tw, tw + n / 2); /* forward r2c :208 */
            /* do_filtering :100-102 -- both components scaled by K[x] (re == im in the reference's K) */
            for(uint32_t x = 0; x < size_trans; ++x)
            {
                re[x] *= k[x];
                im[x] *= k[x];
            }
            /* c2r reads only bins 0..n/2 and assumes Hermitian symmetry; imaginary parts of the DC and
             * Nyquist bins are ignored (FFTW c2r convention) */
            im[0] = 0.f;
            im[n / 2] = 0.f;
            for(uint32_t x = 1; x < n / 2; ++x)
            {
                re[n - x] = re[x];
                im[n - x] = -im[x];
            }
            fft_c2c(re, im, n, +1, tw, tw + n / 2); /* inverse c2r :214 */
            /* shrink :114 + normalize :128 */
            for(uint32_t x = 0; x < dim_x; ++x)
                row[x] = re[x] / (float)filter_size;
        }
        free(re);
        free(im);
    }
    free(tw);
}

/* ------------------------------------------------------------------------------------------------
 * backprojection -- src/backprojection.cpp, src/openmp/backprojection.cpp
 * ---------------------------------------------------------------------------------------------- */

/* src/backprojection.cpp:49-63 */
void po_backproject_constants(const po_detector_geometry* det, uint32_t idx, int enable_angles, float phi_in,
                              float* sin_out, float* cos_out, float* delta_s_mm, float* delta_t_mm)
{
    *delta_s_mm = det->delta_s * det->l_px_row; /* :49 */
    *delta_t_mm = det->delta_t * det->l_px_col; /* :50 */
    float phi = 0.f;
    if(enable_angles)
        phi = phi_in; /* :55 */
    else
        phi = (float)idx * det->delta_phi; /* :57 */
    phi *= (float)M_PI / 180.f; /* :60 */
    *sin_out = sinf(phi); /* :62 */
    *cos_out = cosf(phi); /* :63 */
}

/* src/openmp/backprojection.cpp:39-43 */
static inline float vol_centered_coordinate(uint32_t coord, uint32_t dim, float size)
{
    const float size2 = size / 2.f;
    return -((float)dim * size2) + size2 + (float)coord * size;
}

/* src/openmp/backprojection.cpp:45-50 */
static inline float proj_real_coordinate(float coord, uint32_t dim, float size, float offset)
{
    const float size2 = size / 2.f;
    const float min = -((float)dim * size2) - offset;
    return (coord - min) / size - (1.f / 2.f);
}

/* src/openmp/backprojection.cpp:52-84 */
static inline float interpolate(const float* p, float x, float y, uint32_t dim_x, uint32_t dim_y)
{
    const float x1 = floorf(x);
    const float x2 = x1 + 1.f;
    const float y1 = floorf(y);
    const float y2 = y1 + 1.f;

    const int x1_valid = x1 >= 0.f;
    const int x2_valid = x2 < (float)dim_x;
    const int y1_valid = y1 >= 0.f;
    const int y2_valid = y2 < (float)dim_y;

    float interp = 0.f;
    if(x1_valid && x2_valid && y1_valid && y2_valid)
    {
        const size_t x1u = (size_t)(uint32_t)x1;
        const size_t x2u = (size_t)(uint32_t)x2;
        const size_t y1u = (size_t)(uint32_t)y1;
        const size_t y2u = (size_t)(uint32_t)y2;
        const float q11 = p[x1u + y1u * dim_x];
        const float q12 = p[x1u + y2u * dim_x];
        const float q21 = p[x2u + y1u * dim_x];
        const float q22 = p[x2u + y2u * dim_x];
        const float interp_y1 = (x2 - x) / (x2 - x1) * q11 + (x - x1) / (x2 - x1) * q21; /* :77 */
        const float interp_y2 = (x2 - x) / (x2 - x1) * q12 + (x - x1) / (x2 - x1) * q22; /* :78 */
        interp = (y2 - y) / (y2 - y1) * interp_y1 + (y - y1) / (y2 - y1) * interp_y2;   /* :80 */
    }
    return interp;
}

/* src/openmp/backprojection.cpp:86-153 (do_backprojection) driven by :156-199 (constants) */
void po_backproject(float* vol, uint32_t v_dim_x, uint32_t v_dim_y, uint32_t v_dim_z,
                    const float* p, uint32_t p_dim_x, uint32_t p_dim_y, uint32_t v_offset,
                    const po_detector_geometry* det, const po_volume_geometry* vol_geo,
                    int enable_roi, const po_region_of_interest* roi,
                    float sin_phi, float cos_phi, float delta_s_mm, float delta_t_mm)
{
    const uint32_t v_dim_x_full = vol_geo->dim_x; /* :162-164 */
    const uint32_t v_dim_y_full = vol_geo->dim_y;
    const uint32_t v_dim_z_full = vol_geo->dim_z;
    const float l_vx_x = vol_geo->l_vx_x; /* :166-168 */
    const float l_vx_y = vol_geo->l_vx_y;
    const float l_vx_z = vol_geo->l_vx_z;
    const float l_px_x = det->l_px_row; /* :170-171 */
    const float l_px_y = det->l_px_col;
    const float d_so = det->d_so;                              /* :176 (raw, signed: SURVEY Q12) */
    const float d_sd = fabsf(det->d_so) + fabsf(det->d_od);    /* :177 */
    const uint32_t rx = enable_roi ? roi->x1 : 0u; /* :105-110 */
    const uint32_t ry = enable_roi ? roi->y1 : 0u;
    const uint32_t rz = enable_roi ? roi->z1 : 0u;

    #pragma omp parallel for collapse(2) schedule(static)
    for(uint32_t m = 0; m < v_dim_z; ++m)
    {
        for(uint32_t l = 0; l < v_dim_y; ++l)
        {
            const uint32_t M = m + rz + v_offset; /* :109,:113 */
            const uint32_t L = l + ry;
            const float y_l = vol_centered_coordinate(L, v_dim_y_full, l_vx_y); /* :117 */
            const float z_m = vol_centered_coordinate(M, v_dim_z_full, l_vx_z); /* :118 */
            float* out = vol + ((size_t)l + (size_t)m * v_dim_y) * v_dim_x;     /* :102 */
            for(uint32_t k = 0; k < v_dim_x; ++k)
            {
                const uint32_t K = k + rx;
                const float x_k = vol_centered_coordinate(K, v_dim_x_full, l_vx_x); /* :116 */

                const float s = x_k * cos_phi + y_l * sin_phi;  /* :121 */
                const float t = -x_k * sin_phi + y_l * cos_phi; /* :122 */

                const float factor = d_sd / (s + d_so); /* :125 */
                const float h = proj_real_coordinate(t * factor, p_dim_x, l_px_x, delta_s_mm);   /* :126-129 */
                const float v = proj_real_coordinate(z_m * factor, p_dim_y, l_px_y, delta_t_mm); /* :130-133 */

                const float det_v = interpolate(p, h, v, p_dim_x, p_dim_y); /* :136 */

                const float u = -(d_so / (s + d_so)); /* :139 */
                out[k] += 0.5f * det_v * u * u;       /* :140 */
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * helpers for the known-answer tests
 * ---------------------------------------------------------------------------------------------- */

uint64_t po_fnv1a64(const void* data, size_t n)
{
    const unsigned char* b = (const unsigned char*)data;
    uint64_t h = 0xcbf29ce484222325ull;
    for(size_t i = 0; i < n; ++i)
    {
        h ^= b[i];
        h *= 0x100000001b3ull;
    }
    return h;
}

/* SURVEY.md section 8c input generator: s = 12345 + idx; per pixel s = s*1664525 + 1013904223 (mod 2^32),
 * pixel = (s >> 8) / 2^24 */
void po_lcg_fill(float* p, size_t n, uint32_t idx)
{
    uint32_t s = 12345u + idx;
    for(size_t i = 0; i < n; ++i)
    {
        s = s * 1664525u + 1013904223u;
        p[i] = (float)(s >> 8) / 16777216.f;
    }
}

int po_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void po_set_num_threads(int n)
{
#ifdef _OPENMP
    if(n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
