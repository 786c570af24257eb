/*
 * paris_hip.h -- C ABI of the MI355X (gfx950) backend for the hzdr/PARIS FDK hot path.
 *
 * This is the drop-in boundary. PARIS selects a backend at compile time through a namespace alias
 * (src/backend.h:26-47); a backend is a set of types plus 14 free functions (src/generic/backend.h:54-88,
 * src/openmp/backend.h:42-89, src/cuda/backend.h:50-103). The C++ header paris_amd/host/hip/backend.h
 * implements that surface (namespace paris::hip) on top of the entry points below; INTEGRATION.md shows
 * the `#elif defined(PARIS_ENABLE_HIP)` arm a maintainer adds to src/backend.h.
 *
 * Conventions
 *   - plain pointers and sizes only; every function returns 0 (PARIS_HIP_SUCCESS) or a non-zero status
 *     (hipError_t values are passed through, library-specific ones start at 10000); no exceptions cross
 *     this boundary. paris_hip_strerror() turns a status into text. The C++ wrappers translate a non-zero
 *     status into paris::stage_runtime_error / stage_construction_error (src/exception.h:31-41).
 *   - a paris_hip_ctx replaces every function-local `static` / `thread_local static` of the reference
 *     (src/weighting.cpp:37-42, src/filtering.cpp:37-42, src/backprojection.cpp:49-50,
 *     src/cuda/filtering.cu:189-240, src/cuda/backprojection.cu:160-185): device, stream, FFT scratch.
 *     One ctx per device and host thread; a ctx is not thread-safe (reference contract "thread == device",
 *     src/main.cpp:87,157-167).
 *   - all device work is enqueued on the ctx stream. Calls are asynchronous unless the ctx was created with
 *     PARIS_HIP_CTX_SYNCHRONOUS, in which case every call synchronises the stream before it returns, as
 *     the reference's CUDA backend does (src/cuda/weighting.cu:72, filtering.cu:260, backprojection.cu:236).
 *   - projections are row-major `buf[s + t * pitch/4]`, s in [0, dim_x) fastest (src/openmp/weighting.cpp:41);
 *     `pitch` is the row stride in BYTES (>= dim_x*4, multiple of 4), the same meaning as
 *     cuda::projection_device_buffer_type::pitch() (src/cuda/weighting.cu:43-45).
 *     Volumes are dense `buf[k + l*dim_x + m*dim_x*dim_y]` (src/openmp/backprojection.cpp:102) with 64-bit
 *     indexing (the reference's 32-bit arithmetic, SURVEY.md Q3, is not reproduced).
 */
#ifndef PARIS_HIP_H_
#define PARIS_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PARIS_HIP_SUCCESS 0
#define PARIS_HIP_ERROR_INVALID_ARGUMENT 10001
#define PARIS_HIP_ERROR_NO_DEVICE 10002
#define PARIS_HIP_ERROR_UNSUPPORTED 10003

/* ctx flags */
#define PARIS_HIP_CTX_DEFAULT 0u
#define PARIS_HIP_CTX_SYNCHRONOUS 1u /* sync the stream before every call returns (reference behaviour) */
#define PARIS_HIP_CTX_LEGACY_STREAM 2u /* with stream == NULL: enqueue on the legacy default stream instead of a private one */
/* paris_hip_ctx_create also does the one-off work a per-projection loop would otherwise pay inside its first iterations: the
 * ctx's upload, second and auxiliary streams and their events are created (a stream costs 2-20 ms), the library's code objects are
 * loaded and the runtime's host-to-device path is exercised once. What set_device() of the C++ mirror asks for (src/main.cpp:87
 * binds the device once per thread, before the task loop). Results never depend on it. */
#define PARIS_HIP_CTX_WARM 4u

/* src/geometry.h:30-46, field for field */
typedef struct paris_detector_geometry {
    uint32_t n_row;   /* pixels per row */
    uint32_t n_col;   /* pixels per column */
    float l_px_row;   /* pixel size, horizontal [mm] */
    float l_px_col;   /* pixel size, vertical [mm] */
    float delta_s;    /* horizontal offset [px] */
    float delta_t;    /* vertical offset [px] */
    float d_so;       /* source -> object */
    float d_od;       /* object -> detector */
    float delta_phi;  /* angle step [deg] */
} paris_detector_geometry;

/* src/geometry.h:48-57 */
typedef struct paris_volume_geometry {
    uint32_t dim_x, dim_y, dim_z;
    float l_vx_x, l_vx_y, l_vx_z;
} paris_volume_geometry;

/* src/geometry.h:59-69 */
typedef struct paris_subvolume_geometry {
    uint32_t dim_x, dim_y, dim_z;
    uint32_t remainder;
} paris_subvolume_geometry;

/* src/region_of_interest.h:30-38 */
typedef struct paris_region_of_interest {
    uint32_t x1, x2, y1, y2, z1, z2;
} paris_region_of_interest;

/* src/subvolume_information.h:30-34 */
typedef struct paris_subvolume_info {
    paris_subvolume_geometry geo;
    int num;
} paris_subvolume_info;

typedef struct paris_hip_ctx paris_hip_ctx;

/* ---- device management: backend::get_devices / set_device (src/cuda/device.cpp:31-47,
 *      src/openmp/backend.h:87-89) -------------------------------------------------------------------- */
/* Environment switches (diagnostics, read once per process): PARIS_HIP_VIRTUAL_DEVICES=k reports k device handles
 * mapped round-robin onto the physical GPUs (exercises the one-thread-per-device driver on a one-GPU box);
 * PARIS_HIP_UPLOAD_STREAM=0 keeps paris_hip_upload_projection on the compute stream (A/B of the overlap; experiments build only). */
int paris_hip_device_count(int* count);

/* Creates the per-device state. `stream` is a hipStream_t to enqueue on (e.g. the caller's torch stream),
 * or NULL to let the ctx create and own a non-blocking stream -- which is NOT ordered with work on the legacy
 * default stream: a caller whose own work runs on the default stream (handle 0, e.g. PyTorch without an explicit
 * stream) passes NULL together with PARIS_HIP_CTX_LEGACY_STREAM. Replaces set_device + all thread_local statics. */
int paris_hip_ctx_create(int device, void* stream, unsigned flags, paris_hip_ctx** out);
int paris_hip_ctx_destroy(paris_hip_ctx* ctx);
/* synchronize_stream (src/cuda/stream.cpp) */
int paris_hip_ctx_synchronize(paris_hip_ctx* ctx);
/* the hipStream_t the ctx enqueues on */
void* paris_hip_ctx_stream(paris_hip_ctx* ctx);

/* ---- fences (extension): a marker in the ctx stream the host can wait on, so a pipelined driver can reuse a pinned
 *      upload buffer as soon as the copy that reads it has finished, without draining the whole stream ---------- */
typedef struct paris_hip_fence paris_hip_fence;
int paris_hip_fence_create(paris_hip_ctx* ctx, paris_hip_fence** out);
int paris_hip_fence_record(paris_hip_ctx* ctx, paris_hip_fence* fence);
/* returns at once if never recorded. Touches no ctx state (ctx only names the device): a thread other than the ctx's own may wait for a
 * fence that thread has been told is recorded -- paris.hip's feed and drain threads do (paris_amd/host/paris/reconstruct.h) -- and a fence of
 * one ctx may be waited for through another ctx of the same device. Every other entry point of a ctx belongs to one thread at a time. */
int paris_hip_fence_wait(paris_hip_ctx* ctx, paris_hip_fence* fence);
int paris_hip_fence_destroy(paris_hip_ctx* ctx, paris_hip_fence* fence);

/* ---- memory: make_projection_device / make_volume_device / copy_h2d / copy_d2h
 *      (src/cuda/memory.cpp:33-102, src/openmp/memory.cpp:33-79) ------------------------------------- */
/* device projection, rows padded to a 256-byte multiple; *pitch receives the row stride in bytes */
int paris_hip_malloc_projection(paris_hip_ctx* ctx, uint32_t dim_x, uint32_t dim_y, float** d_ptr, size_t* pitch);
/* device volume, zero-filled (make_volume_device semantics: src/openmp/memory.cpp:46-47) */
int paris_hip_malloc_volume(paris_hip_ctx* ctx, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z, float** d_ptr);
int paris_hip_free(paris_hip_ctx* ctx, void* d_ptr);
/* pinned host memory for projection/volume host buffers (cuda: pinned_host_ptr).
 * Projection-sized buffers (host ones up to 64 MiB) are recycled: paris_hip_free / paris_hip_free_host return at once, and the
 * buffer is handed out again only after the work of THIS API that used it has finished -- a pinned buffer after the last copy from /
 * into it (an upload-stream copy does not wait for anything queued on the ctx stream), a device buffer after everything queued on
 * the ctx stream when it was released. Work the CALLER enqueues itself on a pinned buffer (its own hipMemcpyAsync) is not seen:
 * finish it (paris_hip_ctx_synchronize, a fence) before releasing that buffer. */
int paris_hip_malloc_host(paris_hip_ctx* ctx, size_t bytes, void** h_ptr);
int paris_hip_free_host(paris_hip_ctx* ctx, void* h_ptr);
/* 2-D copies, pitches in bytes; projection rows are dim_x floats */
int paris_hip_memcpy_projection_h2d(paris_hip_ctx* ctx, float* d_dst, size_t d_pitch, const float* h_src,
                                    size_t h_pitch, uint32_t dim_x, uint32_t dim_y);
/* Extension: the same host -> device copy on a dedicated upload stream of the ctx, with the compute stream made to wait
 * for it (hipStreamWaitEvent): the transfer overlaps kernels already queued. Use pinned host memory, and do not
 * refill h_src or overwrite d_dst before the work that reads them has passed a fence; at most 16 uploads in flight. An upload into
 * a buffer of paris_hip_malloc_projection that no call of this API has touched since it was handed out starts at once, whatever the
 * compute stream still holds: work the CALLER enqueued on such a buffer itself (own kernels on the ctx stream) must be complete
 * before the upload -- the library sees only its own entry points. */
int paris_hip_upload_projection(paris_hip_ctx* ctx, float* d_dst, size_t d_pitch, const float* h_src, size_t h_pitch,
                                uint32_t dim_x, uint32_t dim_y);
int paris_hip_memcpy_projection_d2h(paris_hip_ctx* ctx, float* h_dst, size_t h_pitch, const float* d_src,
                                    size_t d_pitch, uint32_t dim_x, uint32_t dim_y);
int paris_hip_memcpy_volume_h2d(paris_hip_ctx* ctx, float* d_dst, const float* h_src, uint32_t dim_x,
                                uint32_t dim_y, uint32_t dim_z);
int paris_hip_memcpy_volume_d2h(paris_hip_ctx* ctx, float* h_dst, const float* d_src, uint32_t dim_x,
                                uint32_t dim_y, uint32_t dim_z);
int paris_hip_memset_volume(paris_hip_ctx* ctx, float* d_ptr, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z);

/* ---- make_subvolume_information (src/cuda/subvolume_information.cpp:63-118) ---------------------- */
/* Splits dim_z into `num` slabs (+ remainder on the last) so that (volume + 10 projections) / devices fits in the free
 * memory of each of `n_devices` devices, doubling the slab count until it does. n_devices <= 0: all visible devices. */
int paris_hip_make_subvolume_information(const paris_volume_geometry* vol_geo,
                                         const paris_detector_geometry* det_geo, int n_devices,
                                         paris_subvolume_info* out);
/* Extension: the same rule for a driver that keeps `reserve_bytes` of its own buffers per device beside the slab (upload
 * slots, half-precision copies, the deferral ring): a slab plus max(reserve_bytes, the reference's 10-projection allowance)
 * must fit. hipErrorOutOfMemory when reserve_bytes alone does not fit a device. */
int paris_hip_make_subvolume_information_reserving(const paris_volume_geometry* vol_geo,
                                                   const paris_detector_geometry* det_geo, int n_devices,
                                                   size_t reserve_bytes, paris_subvolume_info* out);
/* Extension: free and total memory of a device handle (hipMemGetInfo), for sizing driver buffers before planning. */
int paris_hip_device_memory(int device, size_t* free_bytes, size_t* total_bytes);

/* ---- weighting: backend::weight (src/openmp/weighting.cpp:32-57, src/cuda/weighting.cu:62-73) ---- */
/* p[t][s] *= d_sd / sqrt(d_sd^2 + h_s^2 + v_t^2), h_s = l_px_row/2 + s*l_px_row + h_min, v_t likewise.
 * IEEE sqrt and divide: bit-identical to the OpenMP backend. */
int paris_hip_weight(paris_hip_ctx* ctx, float* d_p, size_t pitch, uint32_t dim_x, uint32_t dim_y, float h_min,
                     float v_min, float d_sd, float l_px_row, float l_px_col);

/* Extension (f4): the same weighting restricted to rows [row_first, row_first + row_count) of the dim_y-row projection
 * at d_p; row t keeps its own v_t, so the band is bit-identical to the same rows of a full paris_hip_weight. */
int paris_hip_weight_rows(paris_hip_ctx* ctx, float* d_p, size_t pitch, uint32_t dim_x, uint32_t dim_y,
                          uint32_t row_first, uint32_t row_count, float h_min, float v_min, float d_sd,
                          float l_px_row, float l_px_col);

/* ---- filtering: backend::make_filter / apply_filter (src/openmp/filtering.cpp:139-219,
 *      src/cuda/filtering.cu:172-261) --------------------------------------------------------------- */
/* K = tau * |rFFT_size(r)| with r the band-limited ramp of src/openmp/filtering.cpp:52-73. *d_k receives a
 * device buffer of size/2+1 floats (the reference stores the same value in re and im of a complex; the
 * product is identical) owned by the caller: release with paris_hip_free. size must be a
 * power of two in [8, 16384]. */
int paris_hip_make_filter(paris_hip_ctx* ctx, uint32_t size, float tau, float** d_k);
/* Extension: the same K multiplied by a window. The reference implements the ramp only;
 * PARIS_HIP_WINDOW_SHEPP_LOGAN multiplies bin f by sinc(pi f / size) (1 at DC, 2/pi at Nyquist). */
#define PARIS_HIP_WINDOW_RAMP 0
#define PARIS_HIP_WINDOW_SHEPP_LOGAN 1
int paris_hip_make_filter_windowed(paris_hip_ctx* ctx, uint32_t size, float tau, int window, float** d_k);
/* Selects the window of the K that paris_hip_stage_filter builds and caches (default: the reference's ramp). */
int paris_hip_set_filter_window(paris_hip_ctx* ctx, int window);
/* In place, per detector row: zero-pad to filter_size, FFT, multiply by K, inverse FFT, keep the first dim_x
 * samples, divide by filter_size. n_col is the number of rows (== dim_y; kept because the reference passes
 * it, src/filtering.cpp:44). */
int paris_hip_apply_filter(paris_hip_ctx* ctx, float* d_p, size_t pitch, uint32_t dim_x, uint32_t dim_y,
                           const float* d_k, uint32_t filter_size, uint32_t n_col);

/* Row-filter kernel: 0 = default (filter_size >= 1024 and a K of paris_hip_make_filter*: radix-16 register passes with
 * table twiddles, the kernel that can also weight in its load; radix-2 below 1024), 1 = the radix-2 kernel for every size,
 * 2 = the first radix-16 kernel (twiddles formed per stage; experiments build only: paris_hip_has_experiments, else
 * PARIS_HIP_ERROR_UNSUPPORTED). A K this ctx did not make runs the radix-2 kernel. Same transform; results differ by fp32 rounding only. */
int paris_hip_set_filter_variant(paris_hip_ctx* ctx, int variant);

/* Extension: stage fusion. With enable != 0 a paris_hip_weight / paris_hip_weight_rows call is held back, and the
 * paris_hip_apply_filter call that follows on the same rows runs ONE kernel that applies the weight as it loads the row
 * (same operations, each rounded once: the transform sees the bits the separate weighting would have stored) -- one launch
 * and 8 instead of 16 bytes of memory traffic per pixel for PARIS's unchanged weight(); filter() call pair
 * (src/main.cpp:102-103). Every other entry point that reads, writes, frees or waits for device data first runs a held-back
 * weighting as its own kernel, so results never depend on the switch; only work the CALLER enqueues on the ctx stream between
 * the two calls would see the rows unweighted. Off by default in the C library (reference behaviour call by call), switched
 * on by paris::hip (C++ mirror) and bench.py. Ignored under PARIS_HIP_CTX_SYNCHRONOUS. */
int paris_hip_set_stage_fusion(paris_hip_ctx* ctx, int enable);

/* Extension: tiles no ray reaches. Where none of a projection's rays reaches any voxel column of a wave's tile -- the corners of
 * the grid outside the field of view, slices above / below the cone on the source side: about a tenth of a 2048^3 launch at the
 * natural grid -- the reference adds 0.5 * 0 * u * u = +0 to every voxel (src/openmp/backprojection.cpp:71,140). Adding +0 changes
 * a float only if it is -0, and a volume that paris_hip_malloc_volume allocated (zero-filled) and that nothing but backprojections
 * wrote since cannot hold one (a sum is -0 only if both terms are). For such volumes, with enable != 0 (the default), those waves
 * neither load nor store their tile; the result is bit-identical. Volumes the library did not allocate, and volumes a
 * paris_hip_memcpy_volume_h2d wrote into, always take every addition. */
int paris_hip_set_backproject_skip_invalid(paris_hip_ctx* ctx, int enable);
/* The library learns of writes into a volume only through its own entry points. A caller that writes one some other way -- its own
 * kernel, a torch tensor over the same memory -- and may have stored a -0 says so with paris_hip_volume_mark_dirty: every volume
 * that overlaps [d_ptr, d_ptr + bytes) takes every addition from then on (until a paris_hip_memset_volume covers it whole again).
 * paris_hip_volume_mark_clean is the opposite promise, for ANY device memory, the caller's own included: the range holds no -0
 * right now (freshly zero-filled; written by nothing but backprojections since) and the caller will mark it dirty before writing
 * anything else into it -- and before releasing memory the library did not allocate: the entry is keyed by address and would vouch
 * for whatever is mapped there next. Writing zeros (hipMemset, tensor.zero_()) needs neither call. */
int paris_hip_volume_mark_dirty(paris_hip_ctx* ctx, const void* d_ptr, size_t bytes);
int paris_hip_volume_mark_clean(paris_hip_ctx* ctx, const void* d_ptr, size_t bytes);
/* For memory whose history the caller does not know (a tensor handed in from elsewhere): reads [d_ptr, d_ptr + bytes) once on the
 * device (bytes a multiple of 4, about 5 ms for 32 GiB), stores the number of -0 words found in *negative_zeros (may be NULL) and
 * lists the range as clean when there is none; when there is one the range takes every addition, as if marked dirty. The same
 * duty as after paris_hip_volume_mark_clean holds from then on. Synchronises the ctx stream. */
int paris_hip_volume_scan_clean(paris_hip_ctx* ctx, const void* d_ptr, size_t bytes, uint64_t* negative_zeros);

/* Extension: weighting and row filter of rows [row_first, row_first + row_count) in one launch, explicitly. d_half != NULL:
 * the filtered rows are stored as IEEE half (round to nearest even) into d_half (same row numbering, half_pitch bytes per
 * row) and the fp32 rows stay as they were (BASELINE config 5: saves the conversion pass). d_k must come from
 * paris_hip_make_filter* on this ctx and filter_size must be >= 1024: PARIS_HIP_ERROR_UNSUPPORTED otherwise. */
int paris_hip_weight_filter_rows(paris_hip_ctx* ctx, float* d_p, size_t pitch, uint32_t dim_x, uint32_t dim_y,
                                 uint32_t row_first, uint32_t row_count, float h_min, float v_min, float d_sd,
                                 float l_px_row, float l_px_col, const float* d_k, uint32_t filter_size, uint16_t* d_half,
                                 size_t half_pitch);

/* Extension: the same for a GROUP of n_frames projections in one launch: frame f starts frame_stride bytes behind frame f - 1
 * (>= pitch * dim_y: frames must not overlap; d_half likewise, half_frame_stride bytes apart). For a driver that holds a group of
 * uploaded frames before a paris_hip_backproject_batch: with small detectors a launch per frame is mostly launch latency.
 * Bit-identical to n_frames paris_hip_weight_filter_rows calls. n_frames <= 65535. */
int paris_hip_weight_filter_batch(paris_hip_ctx* ctx, float* d_p, size_t pitch, size_t frame_stride, uint32_t n_frames, uint32_t dim_x,
                                  uint32_t dim_y, uint32_t row_first, uint32_t row_count, float h_min, float v_min, float d_sd,
                                  float l_px_row, float l_px_col, const float* d_k, uint32_t filter_size, uint16_t* d_half,
                                  size_t half_pitch, size_t half_frame_stride);

/* ---- backprojection: backend::backproject (src/openmp/backprojection.cpp:156-199,
 *      src/cuda/backprojection.cu:133-243) ---------------------------------------------------------- */
/* Adds one filtered projection into the (sub)volume d_v of v_dim_x*v_dim_y*v_dim_z voxels whose first
 * slice is global slice v_offset (+ roi->z1 when enable_roi). Voxel-driven, bilinear, zero unless all four
 * neighbours lie inside the detector (OpenMP rule, src/openmp/backprojection.cpp:52-84); every fp32
 * operation is rounded once in the reference's order, so the result is bit-identical to the OpenMP
 * backend's for the same input. sin/cos/delta_s/delta_t are what the wrapper src/backprojection.cpp:49-68
 * derives (delta_* in mm). */
int paris_hip_backproject(paris_hip_ctx* ctx, const float* d_p, size_t p_pitch, uint32_t p_dim_x,
                          uint32_t p_dim_y, float* d_v, uint32_t v_dim_x, uint32_t v_dim_y, uint32_t v_dim_z,
                          uint32_t v_offset, const paris_detector_geometry* det_geo,
                          const paris_volume_geometry* vol_geo, int enable_roi,
                          const paris_region_of_interest* roi, float sin_phi, float cos_phi, float delta_s,
                          float delta_t);

/* Extension (no reference counterpart; BASELINE config 5 "fp16-in / fp32-accum"): the filtered projection is stored
 * as IEEE half (p_pitch in bytes, >= 2*p_dim_x); pixels are widened to fp32 exactly when they are staged and every
 * operation afterwards is the fp32 one of paris_hip_backproject, so the result equals paris_hip_backproject on the
 * half-rounded projection, bit for bit. paris_hip_convert_projection_f16 rounds to nearest even. */
int paris_hip_convert_projection_f16(paris_hip_ctx* ctx, const float* d_src, size_t src_pitch, uint16_t* d_dst,
                                     size_t dst_pitch, uint32_t dim_x, uint32_t dim_y);
int paris_hip_backproject_f16(paris_hip_ctx* ctx, const uint16_t* d_p, size_t p_pitch, uint32_t p_dim_x,
                              uint32_t p_dim_y, float* d_v, uint32_t v_dim_x, uint32_t v_dim_y, uint32_t v_dim_z,
                              uint32_t v_offset, const paris_detector_geometry* det_geo,
                              const paris_volume_geometry* vol_geo, int enable_roi,
                              const paris_region_of_interest* roi, float sin_phi, float cos_phi, float delta_s,
                              float delta_t);

/* Extension (no reference counterpart): backprojects n_proj projections per launch (fused kernel, up to 64 per
 * launch, more are split); projection i is at d_p + i * p_stride_bytes. Every voxel's sum is accumulated in
 * projection order in registers, so the result is bit-identical to n_proj successive paris_hip_backproject calls
 * while the volume is read and written once per launch: 8 / n_proj bytes of HBM traffic per voxel-update. Any volume
 * width and alignment; n_proj == 1 is the single-projection kernel. */
int paris_hip_backproject_batch(paris_hip_ctx* ctx, const float* d_p, size_t p_pitch, size_t p_stride_bytes,
                                uint32_t n_proj, uint32_t p_dim_x, uint32_t p_dim_y, float* d_v,
                                uint32_t v_dim_x, uint32_t v_dim_y, uint32_t v_dim_z, uint32_t v_offset,
                                const paris_detector_geometry* det_geo, const paris_volume_geometry* vol_geo,
                                int enable_roi, const paris_region_of_interest* roi, const float* sin_phi,
                                const float* cos_phi, float delta_s, float delta_t);

/* The same for IEEE-half projections (paris_hip_backproject_f16 semantics; p_pitch / p_stride_bytes in bytes of half rows). */
int paris_hip_backproject_batch_f16(paris_hip_ctx* ctx, const uint16_t* d_p, size_t p_pitch, size_t p_stride_bytes,
                                    uint32_t n_proj, uint32_t p_dim_x, uint32_t p_dim_y, float* d_v, uint32_t v_dim_x,
                                    uint32_t v_dim_y, uint32_t v_dim_z, uint32_t v_offset,
                                    const paris_detector_geometry* det_geo, const paris_volume_geometry* vol_geo,
                                    int enable_roi, const paris_region_of_interest* roi, const float* sin_phi,
                                    const float* cos_phi, float delta_s, float delta_t);

/* Extension: deferred backprojection. With depth n > 1 (n <= 64), paris_hip_backproject (and paris_hip_backproject_f16:
 * half-precision calls form groups of their own) copies its projection into a ring
 * owned by the ctx (stream-ordered device-to-device copy: the caller may reuse its buffer as after any asynchronous
 * call) and returns; the pending projections are added by ONE fused launch, in call order and bit-identical to n
 * single launches, when n are pending, when a call with another volume, slab, geometry or ROI arrives, and before
 * every entry point that observes or changes a volume, completes work or changes how backprojection runs
 * (paris_hip_ctx_synchronize, _fence_record, _memcpy_volume_*, _memset_volume, _free, _backproject_batch, a call of the
 * other precision, the timing and tuning calls, paris_hip_flush). The volume is then read and written once per n projections instead of
 * once per projection. Two caveats: work the caller enqueues on the ctx's stream OUTSIDE this API does not see deferred
 * projections (call paris_hip_flush first), and paris_hip_ctx_destroy runs what is still pending only into a volume that
 * paris_hip_malloc_volume of this ctx allocated and paris_hip_free has not taken back (pending projections of any other volume
 * are dropped: the library cannot know whether that address still belongs to the caller). Depth 1 (the
 * default) is immediate execution. The C++ mirror paris::hip enables depth 48, so PARIS's unchanged per-projection loop
 * (src/main.cpp:98-105) runs at the fused kernel's rate. */
int paris_hip_set_backproject_deferral(paris_hip_ctx* ctx, uint32_t depth);
/* (The first groups of a sequence of calls into one volume are launched early -- after 8, 16 and 32 calls, then every `depth` -- so
 * that the device starts while the caller is still feeding the first full group; a call with other arguments, or a change of the
 * depth, starts a new sequence.) */
int paris_hip_flush(paris_hip_ctx* ctx);
/* Extension: deferral BY REFERENCE. With enable != 0 a deferred paris_hip_backproject[_f16] whose projection is a whole buffer of
 * paris_hip_malloc_projection (the pointer it returned, its pitch) takes NO snapshot: the group's fused launch reads the buffer
 * itself. The library answers for such a buffer until that launch has been made: paris_hip_free of it returns at once and the
 * buffer goes back to the pool behind the launch (one event per group instead of one per buffer, no copy enqueued per call); any
 * other entry point that reads or writes the buffer -- a second weighting or filter, an upload into it, a copy to the host --
 * launches the pending group first, so every result is what the snapshot would have given, bit for bit. With filter deferral
 * the held-back weight + filter of such a projection runs IN PLACE in the group's one filter launch: the buffer holds the filtered
 * pixels whenever anything looks at it through this API. The one thing the library cannot see is work the caller enqueues on
 * the ctx stream by itself: a caller whose own kernels write projection buffers after backprojecting them leaves this off (the
 * default; snapshots). Projections in memory the library did not allocate, row-band pointers into a buffer, other pitches and
 * calls made while a caller-supplied ctx stream is being captured into a graph are snapshotted as before (a group may mix both).
 * A caller that REFILLS one buffer for every projection gains nothing from it -- each refill finds the buffer referenced and launches
 * the pending group, one projection per launch -- and keeps the snapshots. paris::hip switches it on: PARIS's loop
 * (src/main.cpp:98-105) allocates, fills, backprojects and frees one buffer per projection (src/loader.cpp:28-33) and touches
 * nothing outside the backend. */
int paris_hip_set_backproject_references(paris_hip_ctx* ctx, int enable);
/* What this ctx may keep allocated for projections of dim_x x dim_y pixels beside the volume, with its present deferral settings (the
 * rotation of paris_hip_malloc_projection buffers, the pending group's buffers or the snapshot ring): the reserve_bytes a driver
 * passes to paris_hip_make_subvolume_information_reserving. */
int paris_hip_projection_reserve_bytes(paris_hip_ctx* ctx, uint32_t dim_x, uint32_t dim_y, size_t* bytes);
/* How many projections are pending right now and into which volume (NULL when none): what a wrapper that lends the library memory
 * it does not own needs to decide whether a last paris_hip_flush is still safe (paris_amd.backend.Backend.close). */
int paris_hip_pending_backprojections(paris_hip_ctx* ctx, uint32_t* count, void** d_v);
/* Extension: filter deferral, for small projections whose weight + filter launch is mostly latency (512^2: 7 us alone, 0.9 us as one
 * of 48 in a group launch). With enable != 0, stage fusion on and a deferral depth > 1, the paris_hip_apply_filter call that
 * follows a held-back weighting of the same rows is held back as well. If the NEXT call backprojects that projection
 * (paris_hip_backproject into the deferral ring), the library snapshots the still unfiltered frame and runs weighting + filter on
 * its snapshots, one launch for the whole group, right before the group's fused backprojection: the volume is bit-identical. Any
 * other entry point runs the held-back launch first, in place, as if it had never been held. The one thing that changes: after
 * such a backprojection the caller's projection buffer still holds the UNFILTERED pixels (PARIS's loop, src/main.cpp:98-105,
 * never looks at a projection again after backprojecting it; a caller that does leaves this off). Off by default in the bare
 * library. enable == 2 holds a filter back ONLY where nobody can tell: with deferral by reference on
 * (paris_hip_set_backproject_references) and a projection that is a whole buffer of paris_hip_malloc_projection -- the filter then
 * runs IN PLACE in the group's one filter launch and every other call that touches the buffer runs it first, so the buffer holds
 * the filtered pixels whenever anything looks at it through this API; everything else is filtered at once. That is what the C++
 * mirror paris::hip switches on (macro PARIS_HIP_FILTER_DEFERRAL = 2): one filter launch per group instead of one per projection
 * (config 1 through PARIS's loop: profiles/r05_demo_paris_hip_mirror.txt). */
int paris_hip_set_filter_deferral(paris_hip_ctx* ctx, int enable);
/* Extension: asynchronous validation. The hand-expanded IEEE sequences (fast division by the pixel pitch, shared-reciprocal
 * divisions, short sqrt / divide of the weighting) are used only after a device validator has proved them for the operands at hand,
 * once per process, device and operand range; by default the call that needs an answer first waits for the validator (2.3 ms for
 * the exhaustive check of a pixel pitch). With enable != 0 that call launches the validator on the ctx's auxiliary stream and
 * goes on with the compiler's IEEE forms -- the same bits, a few instructions more per voxel column -- until a later call finds
 * the answer there. Results never depend on it; paris_hip_fast_division_is_exact and the paris_hip_lean_*_is_exact queries always
 * wait. Off by default in the bare library (which kernel form a given launch uses then depends on timing), on in paris::hip. */
int paris_hip_set_async_validation(paris_hip_ctx* ctx, int enable);
/* Where the fused launch of a full group runs: with enable != 0 on a second stream of the ctx, ordered behind the
 * group's snapshot copies, so that the uploads, weightings and filters of the NEXT group -- which the caller keeps enqueuing on
 * the ctx stream -- execute beside it instead of behind it (what small volumes need: a 256^3 launch of 16 projections takes about
 * as long as the sixteen copy + filter launches of the next group). Every entry point that flushes (see above) also makes the ctx
 * stream wait for the launches on the second stream, so the caller sees one stream's worth of ordering. Not used under
 * PARIS_HIP_CTX_SYNCHRONOUS or while the ctx stream is being captured into a graph. Results never depend on it. Off by default in
 * the bare library; the C++ mirror paris::hip switches it on (macro PARIS_HIP_BACKPROJECT_OVERLAP): PARIS's loop uploads, filters and
 * snapshots the next group while a launch runs, and without the second stream all of that -- and the release of every buffer of
 * the loop -- queues behind the launch: whole circles through the mirror 1.65 -> 1.97 TVox/s at 2048^2, 1.30 -> 1.90 at 1024^2,
 * 0.88 -> 1.78 at 512^2 (profiles/r04_demo_paris_hip_mirror.txt; a call with the same arguments as the group launched last
 * continues without joining the streams -- until round 4 every group's first call joined, and nothing overlapped). With the
 * projections already resident on the device there is nothing to pass and the switch changes nothing (bench.py --overlap). */
int paris_hip_set_backproject_overlap(paris_hip_ctx* ctx, int enable);

/* ---- stage wrappers and geometry (host code of the hot path) ----------------------------------------
 * The reference derives the kernels' scalar arguments in backend-neutral wrappers and caches them in
 * function-local statics; these entry points restate them per call (no statics). */
/* calculate_volume_geometry (src/geometry.cpp:36-84) */
int paris_hip_calculate_volume_geometry(const paris_detector_geometry* det_geo, paris_volume_geometry* out);
/* apply_roi (src/geometry.cpp:86-130); an invalid ROI returns the input geometry, as the reference does */
int paris_hip_apply_roi(const paris_volume_geometry* vol_geo, const paris_region_of_interest* roi,
                        paris_volume_geometry* out);
/* paris::weight (src/weighting.cpp:32-45): derives h_min, v_min, d_sd and calls paris_hip_weight */
int paris_hip_stage_weight(paris_hip_ctx* ctx, float* d_p, size_t pitch, uint32_t dim_x, uint32_t dim_y,
                           const paris_detector_geometry* det_geo);
/* filter_size = 2 * 2^ceil(log2(n_row)) (src/filtering.cpp:37) */
uint32_t paris_hip_filter_size(uint32_t n_row);
/* paris::filter (src/filtering.cpp:32-45): builds K once per ctx, then paris_hip_apply_filter */
int paris_hip_stage_filter(paris_hip_ctx* ctx, float* d_p, size_t pitch, uint32_t dim_x, uint32_t dim_y,
                           const paris_detector_geometry* det_geo);
/* Extensions (f4): the two wrappers above on rows [row_first, row_first + row_count) of the projection at d_p only. With a
 * band from paris_hip_slab_row_band (whole filter row pairs) the band's pixels are bit-identical to a full call. */
int paris_hip_stage_weight_rows(paris_hip_ctx* ctx, float* d_p, size_t pitch, uint32_t dim_x, uint32_t dim_y,
                                uint32_t row_first, uint32_t row_count, const paris_detector_geometry* det_geo);
int paris_hip_stage_filter_rows(paris_hip_ctx* ctx, float* d_p, size_t pitch, uint32_t dim_x, uint32_t dim_y,
                                uint32_t row_first, uint32_t row_count, const paris_detector_geometry* det_geo);
/* Extension: paris::weight + paris::filter of a row band in one launch (paris_hip_weight_filter_rows with the wrappers'
 * constants and cached K); d_half != NULL stores the band as IEEE half there (half_pitch bytes per row, same row numbering)
 * and leaves the fp32 rows unfiltered. Narrow detectors (filter length < 1024) run the separate stages: same result. */
int paris_hip_stage_weight_filter_rows(paris_hip_ctx* ctx, float* d_p, size_t pitch, uint32_t dim_x, uint32_t dim_y,
                                       uint32_t row_first, uint32_t row_count, const paris_detector_geometry* det_geo,
                                       uint16_t* d_half, size_t half_pitch);
/* Extension: paris_hip_stage_weight_filter_rows for a group of n_frames projections frame_stride bytes apart in one launch
 * (paris_hip_weight_filter_batch with the wrappers' constants and cached K; narrow detectors run frame by frame: same result). */
int paris_hip_stage_weight_filter_batch(paris_hip_ctx* ctx, float* d_p, size_t pitch, size_t frame_stride, uint32_t n_frames,
                                        uint32_t dim_x, uint32_t dim_y, uint32_t row_first, uint32_t row_count,
                                        const paris_detector_geometry* det_geo, uint16_t* d_half, size_t half_pitch,
                                        size_t half_frame_stride);
/* Extension (f4; no reference counterpart): the detector rows [*row_first, *row_first +
 * *row_count) that backprojecting into the slab (v_dim_*, v_offset, optional ROI; arguments as paris_hip_backproject)
 * can read for ANY projection angle. Rows outside the band never contribute to the slab, so a driver may upload,
 * weight (paris_hip_weight_rows) and filter (paris_hip_apply_filter on the band's rows) only the band and still get
 * the bit-identical volume; the projection buffer keeps its full size. The band starts on an even row and ends on an
 * odd one (or the last row): the filter transforms row pairs. Conservative: the whole detector when no bound
 * exists (source distance inside the slab's circle), *row_count = 0 when the slab never projects onto the detector. */
int paris_hip_slab_row_band(const paris_detector_geometry* det_geo, const paris_volume_geometry* vol_geo,
                            uint32_t v_dim_x, uint32_t v_dim_y, uint32_t v_dim_z, uint32_t v_offset, int enable_roi,
                            const paris_region_of_interest* roi, uint32_t* row_first, uint32_t* row_count);
/* angle of projection idx -> sin/cos on the host in fp32 (src/backprojection.cpp:52-63) */
int paris_hip_stage_angle(const paris_detector_geometry* det_geo, uint32_t idx, int enable_angles, float phi,
                          float* sin_phi, float* cos_phi);
/* paris::backproject (src/backprojection.cpp:37-69): p_idx / p_phi are projection::idx / projection::phi */
int paris_hip_stage_backproject(paris_hip_ctx* ctx, const float* d_p, size_t p_pitch, uint32_t p_dim_x,
                                uint32_t p_dim_y, uint32_t p_idx, float p_phi, float* d_v, uint32_t v_dim_x,
                                uint32_t v_dim_y, uint32_t v_dim_z, uint32_t v_offset,
                                const paris_detector_geometry* det_geo, const paris_volume_geometry* vol_geo,
                                int enable_angles, int enable_roi, const paris_region_of_interest* roi);

/* ---- diagnostics ---------------------------------------------------------------------------------- */
const char* paris_hip_strerror(int status);
/* library version "major.minor.patch" */
const char* paris_hip_version(void);
/* 1: this library is the experiments build (make -C paris_amd/csrc EXPERIMENTS=1 -> libparis_hip_experiments.so): the kernels,
 * tile orders and A/B switches that were measured and lost are compiled in. 0: the product build -- its setters answer
 * PARIS_HIP_ERROR_UNSUPPORTED for them (backprojection variants 3 and 5, unroll 3 and 4, the slice shapes, tile orders 0 / 1 / 8 /
 * 9 / 12, filter variant 2). Results never depend on any of them. */
int paris_hip_has_experiments(void);
/* Times the most recent backproject launch on this ctx (HIP events recorded on the ctx stream around the
 * kernel); synchronises the stream. Used by bench.py for roofline.achieved. */
int paris_hip_last_backproject_ms(paris_hip_ctx* ctx, float* ms);
/* Arms a ring of `capacity` HIP event pairs: every following backproject launch on this ctx is bracketed by
 * events on the ctx stream. _collect synchronises the stream and returns the durations (ms) of the launches
 * still in the ring, oldest first. Default capacity is 1 (paris_hip_last_backproject_ms). Capacity 0 switches the events
 * off (nothing but kernels and copies is enqueued: what a caller capturing the ctx stream into a hipGraph wants). */
int paris_hip_backproject_timing_arm(paris_hip_ctx* ctx, uint32_t capacity);
int paris_hip_backproject_timing_collect(paris_hip_ctx* ctx, float* ms, uint32_t max_n, uint32_t* n_out);
/* Selects the backprojection kernel: 0 = default (currently the tile kernel), 1 = one-thread-per-voxel gather kernel
 * without LDS (slow, for cross-checking), 2 = tile kernel (z-walk per workgroup), 3 = slice kernel (one slice per
 * wave; needs a 16-byte aligned volume with dim_x % 4 == 0, else falls back to 2), 4 = the fused kernel run with one
 * projection (every slice of a tile in flight before the first store; measured 5 % slower than the tile kernel). All give
 * identical bits. 3 and 5 (the two-pass variant: column constants of the whole plane precomputed per projection) were measured
 * slower and exist in the experiments build only (PARIS_HIP_ERROR_UNSUPPORTED in the product).
 * paris_hip_backproject_batch uses its fused kernel under variant 0 and runs one launch per projection otherwise. */
int paris_hip_set_backproject_variant(paris_hip_ctx* ctx, int variant);
/* Tuning knobs of the LDS-staged kernel; 0 keeps the default. vx: voxels per lane along x (1, 2, 4; capped by
 * the volume's alignment), unroll: z slices in flight per lane (1, 2, 4; 3 = one at a time with the next one prefetched), tz: slices per tile, lds_bytes: LDS
 * budget per workgroup for the staged detector box (1024..65536). Results do not depend on any of them. The launcher picks unroll 1 or 2
 * by itself; 3 and 4 are compiled into the experiments build only (PARIS_HIP_ERROR_UNSUPPORTED in the product). */
int paris_hip_set_backproject_tuning(paris_hip_ctx* ctx, int vx, int unroll, int tz, int lds_bytes);
/* Shape of the slice kernel: waves (= slices per tile) x row groups per lane; (16,4) (16,2) (8,4) (8,2) (8,1),
 * (0,0) = default. Experiments build only (the product has no slice kernel: PARIS_HIP_ERROR_UNSUPPORTED for anything but (0,0)). */
int paris_hip_set_backproject_slice_shape(paris_hip_ctx* ctx, int waves, int row_groups);
/* Workgroup -> tile order (-1 default by volume shape: 15 for planes beyond 1024^2, else 18, or 5 when the z tiles do not divide among the 8 XCDs; 0 x-fastest, 1 z-fastest, 5 one
 * contiguous run of tiles per XCD, 8 one band of y tiles per XCD swept x -> z -> y, 9 the same swept x -> y -> z, 12 order 8 in chunks
 * of 256 slices, 14 / 15 / 16 / 17 y tiles DEALT to the XCDs singly / in pairs / in fours / in eights, in shallow z chunks, 18 z tiles
 * dealt to the XCDs) and the cache
 * policy of the volume stream (2 nontemporal loads with write-through nontemporal stores, 1 nontemporal, 0 plain, -1 library
 * default: plain for slabs that largely stay in the Infinity Cache between launches, up to 384 MiB, 2 for larger ones).
 * Performance only. The product build has the orders the launcher picks from, 5 and 14 .. 18; 0 / 1 / 8 / 9 / 12 lost and are
 * compiled into the experiments build only (PARIS_HIP_ERROR_UNSUPPORTED). */
int paris_hip_set_backproject_order(paris_hip_ctx* ctx, int order, int nontemporal);
/* The per-voxel division by the detector pixel pitch may run as multiply + 2 FMA instead of the IEEE sequence,
 * but only for a divisor for which an exhaustive GPU check over all 2^32 fp32 dividends (once per ctx and divisor)
 * found the detector coordinate v = x / pitch - 0.5 identical (bit for bit, or out of any detector's range both ways).
 * enable = 0 forces the IEEE sequence. Results never depend on this. */
int paris_hip_set_backproject_fast_division(paris_hip_ctx* ctx, int enable);
/* Staging of the detector box into LDS 4 pixels per lane (default on; used when the projection base and pitch are
 * 16-byte aligned, 8-byte for half pixels). enable = 0 forces one pixel per lane. Results never depend on this. */
int paris_hip_set_backproject_vector_staging(paris_hip_ctx* ctx, int enable);
/* Runs (or looks up) the exhaustive check for one divisor: *exact = 1 when multiply + 2 FMA reproduces
 * x / divisor - 0.5 for every fp32 x (see above). */
int paris_hip_fast_division_is_exact(paris_hip_ctx* ctx, float divisor, int* exact);
/* Two more hand-expanded IEEE sequences, each used only after the device has compared it with the compiler's correctly rounded
 * operation for EVERY fp32 operand a launch can produce (once per process, device and operand range; microseconds):
 *  - the two per-column divisions d_sd / (s + d_so), d_so / (s + d_so) of the backprojection (src/openmp/backprojection.cpp:125,139)
 *    share one refined reciprocal; checked for every denominator in [0.09, 1.92] x d_so, the range the library first proves the
 *    launch stays in (otherwise, or on any mismatch, both are plain IEEE divisions). *exact = 1: both quotients have the IEEE bits.
 *  - the weighting's d_sd / sqrt(q) (src/openmp/weighting.cpp:52) inside the one-launch weight + filter; checked for every radicand
 *    q in [q_lo, q_hi] (the library asks for [d_sd^2, largest radicand of the rows], widened a little).
 * paris_hip_set_lean_validation(ctx, 0) makes the kernels trust the range tests alone (rounds 2-3 behaviour; for A/B and for testing
 * the validators). Results never depend on any of this. */
int paris_hip_lean_division_is_exact(paris_hip_ctx* ctx, float d_sd, float d_so, int* exact);
int paris_hip_lean_weighting_is_exact(paris_hip_ctx* ctx, float d_sd, float q_lo, float q_hi, int* exact);
int paris_hip_set_lean_validation(paris_hip_ctx* ctx, int enable);

#ifdef __cplusplus
}
#endif
#endif /* PARIS_HIP_H_ */
