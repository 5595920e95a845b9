// hip_api.hip -- implementation of include/crpspmm_hip.h (device-level C ABI).
// Replaces the host/CUDA shims of /root/reference/deprecated/src/cuda_proxy.cu:53-182
// with HIP-only code for gfx950; there is no CPU fallback in this file.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include "crpspmm_hip.h"
#include "kernels.h"

struct crp_csr_dev
{
    int       nrow;
    int       ncol;
    long long nnz;
    int      *rowptr;
    int      *colidx;
    double   *val;
};

#define CRP_TRY(expr)                                 \
    do                                                \
    {                                                 \
        hipError_t e__ = (expr);                      \
        if (e__ != hipSuccess) return (int) e__;      \
    } while (0)

extern "C" {

const char *crp_hip_version(void) { return "crpspmm-hip 0.1 gfx950"; }

int crp_hip_device_count(int *count)
{
    if (count == NULL) return -1;
    *count = 0;
    CRP_TRY(hipGetDeviceCount(count));
    return 0;
}

int crp_hip_set_device(int dev) { CRP_TRY(hipSetDevice(dev)); return 0; }
int crp_hip_get_device(int *dev) { if (!dev) return -1; CRP_TRY(hipGetDevice(dev)); return 0; }

int crp_hip_device_info(int dev, char *name, int *cu_count, size_t *hbm_bytes)
{
    hipDeviceProp_t prop;
    CRP_TRY(hipGetDeviceProperties(&prop, dev));
    if (name)
    {
        snprintf(name, 256, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    return 0;
}

int crp_dev_malloc(void **ptr, size_t bytes)
{
    if (ptr == NULL) return -1;
    *ptr = NULL;
    if (bytes == 0) return 0;
    CRP_TRY(hipMalloc(ptr, bytes));
    return 0;
}

int crp_dev_free(void *ptr)
{
    if (ptr == NULL) return 0;
    CRP_TRY(hipFree(ptr));
    return 0;
}

int crp_dev_memset(void *ptr, int value, size_t bytes, void *stream)
{
    if (bytes == 0) return 0;
    CRP_TRY(hipMemsetAsync(ptr, value, bytes, (hipStream_t) stream));
    return 0;
}

int crp_dev_memcpy(void *dst, const void *src, size_t bytes, int kind, void *stream)
{
    if (bytes == 0) return 0;
    hipMemcpyKind k;
    if (kind == 0) k = hipMemcpyHostToDevice;
    else if (kind == 1) k = hipMemcpyDeviceToHost;
    else if (kind == 2) k = hipMemcpyDeviceToDevice;
    else return -1;
    CRP_TRY(hipMemcpyAsync(dst, src, bytes, k, (hipStream_t) stream));
    return 0;
}

int crp_dev_ptr_is_device(const void *ptr, int *is_dev)
{
    if (is_dev == NULL) return -1;
    *is_dev = 0;
    if (ptr == NULL) return 0;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, ptr);
    if (e != hipSuccess)
    {
        (void) hipGetLastError();   // plain malloc'd host memory is "invalid value": not an error for us
        return 0;
    }
    *is_dev = (attr.type == hipMemoryTypeDevice) ? 1 : 0;
    return 0;
}

int crp_stream_create(void **stream)
{
    if (stream == NULL) return -1;
    hipStream_t s;
    CRP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *) s;
    return 0;
}
int crp_stream_destroy(void *stream) { if (stream) CRP_TRY(hipStreamDestroy((hipStream_t) stream)); return 0; }
int crp_stream_sync(void *stream) { CRP_TRY(hipStreamSynchronize((hipStream_t) stream)); return 0; }

int crp_event_create(void **event)
{
    if (event == NULL) return -1;
    hipEvent_t e;
    CRP_TRY(hipEventCreate(&e));
    *event = (void *) e;
    return 0;
}
int crp_event_destroy(void *event) { if (event) CRP_TRY(hipEventDestroy((hipEvent_t) event)); return 0; }
int crp_event_record(void *event, void *stream) { CRP_TRY(hipEventRecord((hipEvent_t) event, (hipStream_t) stream)); return 0; }
int crp_event_sync(void *event) { CRP_TRY(hipEventSynchronize((hipEvent_t) event)); return 0; }
int crp_stream_wait_event(void *stream, void *event)
{
    CRP_TRY(hipStreamWaitEvent((hipStream_t) stream, (hipEvent_t) event, 0));
    return 0;
}
int crp_event_elapsed_ms(void *start, void *stop, float *ms)
{
    if (ms == NULL) return -1;
    CRP_TRY(hipEventElapsedTime(ms, (hipEvent_t) start, (hipEvent_t) stop));
    return 0;
}

// ---------------------------------------------------------------------------
int crp_csr_dev_create(int nrow, int ncol, const int *rowptr, const int *colidx, const double *val,
                       crp_csr_dev_p *out)
{
    if (out == NULL) return -1;
    *out = NULL;
    if (nrow < 0 || ncol < 0 || rowptr == NULL) return -1;
    if (rowptr[0] != 0) return -2;
    const long long nnz = rowptr[nrow];
    if (nnz < 0) return -2;
    if (nnz > 0 && (colidx == NULL || val == NULL)) return -1;
    crp_csr_dev *A = new (std::nothrow) crp_csr_dev;
    if (A == NULL) return -3;
    memset(A, 0, sizeof(*A));
    A->nrow = nrow;
    A->ncol = ncol;
    A->nnz  = nnz;
    hipError_t e;
    e = hipMalloc((void **) &A->rowptr, sizeof(int) * ((size_t) nrow + 1));
    if (e == hipSuccess) e = hipMalloc((void **) &A->colidx, sizeof(int) * (size_t) (nnz > 0 ? nnz : 1));
    if (e == hipSuccess) e = hipMalloc((void **) &A->val, sizeof(double) * (size_t) (nnz > 0 ? nnz : 1));
    if (e == hipSuccess) e = hipMemcpy(A->rowptr, rowptr, sizeof(int) * ((size_t) nrow + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess && nnz > 0) e = hipMemcpy(A->colidx, colidx, sizeof(int) * (size_t) nnz, hipMemcpyHostToDevice);
    if (e == hipSuccess && nnz > 0) e = hipMemcpy(A->val, val, sizeof(double) * (size_t) nnz, hipMemcpyHostToDevice);
    if (e != hipSuccess)
    {
        crp_csr_dev_p tmp = A;
        crp_csr_dev_destroy(&tmp);
        return (int) e;
    }
    *out = A;
    return 0;
}

int crp_csr_dev_destroy(crp_csr_dev_p *A_)
{
    if (A_ == NULL || *A_ == NULL) return 0;
    crp_csr_dev *A = *A_;
    if (A->rowptr) (void) hipFree(A->rowptr);
    if (A->colidx) (void) hipFree(A->colidx);
    if (A->val) (void) hipFree(A->val);
    delete A;
    *A_ = NULL;
    return 0;
}

int crp_csr_dev_nrow(crp_csr_dev_p A) { return A ? A->nrow : -1; }
long long crp_csr_dev_nnz(crp_csr_dev_p A) { return A ? A->nnz : -1; }
long long crp_csr_dev_bytes(crp_csr_dev_p A) { return A ? 12LL * A->nnz + 4LL * ((long long) A->nrow + 1) : -1; }

static const char *k_variant_names[] = {"auto", "rowgroup"};
int crp_spmm_variant_count(void) { return (int) (sizeof(k_variant_names) / sizeof(k_variant_names[0])); }
const char *crp_spmm_variant_name(int variant)
{
    if (variant < 0 || variant >= crp_spmm_variant_count()) return NULL;
    return k_variant_names[variant];
}

int crp_spmm_csr_f64(crp_csr_dev_p A, int layout, int n, const double *B0, long long ldB0, const double *B1,
                     long long ldB1, double *C, long long ldC, int variant, void *stream)
{
    if (A == NULL || n < 0) return -1;
    if (layout != CRP_LAYOUT_ROW_MAJOR && layout != CRP_LAYOUT_COL_MAJOR) return -1;
    if (variant < 0 || variant >= crp_spmm_variant_count()) return -1;
    if (A->nrow == 0 || n == 0) return 0;
    if (C == NULL || (B0 == NULL && B1 == NULL && A->nnz > 0)) return -1;
    if (layout == CRP_LAYOUT_ROW_MAJOR && (ldC < n || (B0 && ldB0 < n) || (B1 && ldB1 < n))) return -4;
    if (layout == CRP_LAYOUT_COL_MAJOR && ldC < A->nrow) return -4;
    crp::SpmmArgs a;
    a.nrow = A->nrow; a.n = n;
    a.rowptr = A->rowptr; a.colidx = A->colidx; a.val = A->val;
    a.B0 = B0; a.ldB0 = ldB0; a.B1 = B1; a.ldB1 = ldB1; a.C = C; a.ldC = ldC;
    hipError_t e;
    if (layout == CRP_LAYOUT_COL_MAJOR) e = crp::spmm_cm_f64(a, (hipStream_t) stream);
    else e = crp::spmm_rm_f64_rowgroup(a, (hipStream_t) stream);
    return (int) e;
}

int crp_gather_rows_f64(int layout, int nidx, int n, const int *ridx, const double *src, long long lds,
                        double *dst, long long ldd, void *stream)
{
    if (nidx < 0 || n < 0 || (layout != 0 && layout != 1)) return -1;
    return (int) crp::gather_rows_f64(layout, nidx, n, ridx, src, lds, dst, ldd, (hipStream_t) stream);
}

int crp_scatter_rows_f64(int layout, int nidx, int n, const int *ridx, const double *src, long long lds,
                         double *dst, long long ldd, void *stream)
{
    if (nidx < 0 || n < 0 || (layout != 0 && layout != 1)) return -1;
    return (int) crp::scatter_rows_f64(layout, nidx, n, ridx, src, lds, dst, ldd, (hipStream_t) stream);
}

int crp_transpose_f64(int nrow, int ncol, const double *src, long long lds, double *dst, long long ldd,
                      void *stream)
{
    if (nrow < 0 || ncol < 0) return -1;
    return (int) crp::transpose_f64(nrow, ncol, src, lds, dst, ldd, (hipStream_t) stream);
}

}  // extern "C"
