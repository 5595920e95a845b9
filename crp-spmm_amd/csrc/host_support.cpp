// host_support.cpp -- the small host-side services behind include/utils.h and include/dev_type.h:
// wall clock, even block splits, norms, strided copies, and the host / device memory-space
// dispatch of the redistribution engine.  Entry points and their observable behaviour are the
// reference's (/root/reference/src/utils.h:104-191, src/dev_type.h:21-57); device memory goes
// through the C ABI of this library (crpspmm_hip.h), pinned host memory when a GPU runtime is up.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "crpspmm_hip.h"
#include "dev_type.h"
#include "par.h"
#include "utils.h"

namespace {

enum class Space { Bad, Host, Device };

Space space_of(dev_type_t t)
{
    switch (t)
    {
        case DEV_TYPE_HOST: return Space::Host;
        case DEV_TYPE_HIP:
        case DEV_TYPE_HIP_RCCL: return Space::Device;
        default: return Space::Bad;
    }
}

bool reject(dev_type_t t)
{
    if (space_of(t) != Space::Bad) return false;
    ERROR_PRINTF("Invalid device type %d\n", t);
    return true;
}

// sum of squares of x, and of x - y when y is given
template <bool WITH_DIFF>
void squares(long long n, const double *x, const double *y, double *sx, double *sd)
{
    double a = 0.0, b = 0.0;
    for (long long i = 0; i < n; i++)
    {
        a += x[i] * x[i];
        if (WITH_DIFF)
        {
            const double d = x[i] - y[i];
            b += d * d;
        }
    }
    *sx = a;
    if (WITH_DIFF) *sd = b;
}

}  // namespace

extern "C" {

// ---- utils.h ---------------------------------------------------------------------------------
double get_wtime_sec(void)
{
    using clk = std::chrono::system_clock;
    return std::chrono::duration<double>(clk::now().time_since_epoch()).count();
}

// len elements in nblk blocks, the remainder spread over the leading blocks; iblk == nblk is the end
// sentinel (start = len; its "size" comes out as len / nblk, as in the reference), anything outside
// [0, nblk] gives (-1, 0)
void calc_block_spos_size(const int len, const int nblk, const int iblk, int *blk_spos, int *blk_size)
{
    *blk_spos = -1;
    *blk_size = 0;
    if (iblk < 0 || iblk > nblk) return;
    const int base = len / nblk, extra = len % nblk;
    const int lead = iblk < extra ? iblk : extra;          // blocks before iblk that carry one more
    *blk_spos = iblk * base + lead;
    *blk_size = base + (iblk < extra ? 1 : 0);
}

void *malloc_aligned(size_t size, size_t alignment)
{
    void *mem = NULL;
    return posix_memalign(&mem, alignment, size) == 0 ? mem : NULL;
}

void free_aligned(void *mem) { free(mem); }

double calc_2norm(const int len, const double *x)
{
    double s = 0.0;
    squares<false>(len, x, NULL, &s, NULL);
    return std::sqrt(s);
}

void calc_err_2norm(const int len, const double *x0, const double *x1, double *x0_2norm_, double *err_2norm_)
{
    double s0 = 0.0, sd = 0.0;
    squares<true>(len, x0, x1, &s0, &sd);
    *x0_2norm_ = std::sqrt(s0);
    *err_2norm_ = std::sqrt(sd);
}

void copy_matrix(const size_t dt_size, const int nrow, const int ncol, const void *src, const int lds,
                 void *dst, const int ldd, const int use_omp)
{
    const size_t row_bytes = dt_size * (size_t) ncol, src_pitch = dt_size * (size_t) lds, dst_pitch = dt_size * (size_t) ldd;
    const char *from = static_cast<const char *>(src);
    char *to = static_cast<char *>(dst);
    auto band = [=](long long first, long long last, int) {
        for (long long r = first; r < last; r++) memcpy(to + (size_t) r * dst_pitch, from + (size_t) r * src_pitch, row_bytes);
    };
    if (use_omp) crp::parallel_chunks(nrow, 1024, band);
    else band(0, nrow, 0);
}

void print_matrix(const int dtype, const int stype, const void *mat, const int ldm, const int nrow,
                  const int ncol, const char *fmt, const char *name)
{
    // stype 0: row-major (element (i, j) at i * ldm + j), otherwise column-major; dtype 0 int, 1 double
    const size_t step_i = stype == 0 ? (size_t) ldm : 1, step_j = stype == 0 ? 1 : (size_t) ldm;
    printf("%s:\n", name);
    for (int i = 0; i < nrow; i++, printf("\n"))
        for (int j = 0; j < ncol; j++)
        {
            const size_t at = i * step_i + j * step_j;
            if (dtype == 0) printf(fmt, static_cast<const int *>(mat)[at]);
            else if (dtype == 1) printf(fmt, static_cast<const double *>(mat)[at]);
        }
}

void dump_binary(const char *fname, void *data, const size_t bytes)
{
    if (FILE *out = fopen(fname, "wb"))
    {
        fwrite(data, 1, bytes, out);
        fclose(out);
    }
}

// ---- dev_type.h --------------------------------------------------------------------------------
int is_dev_type_valid(dev_type_t dev_type) { return space_of(dev_type) != Space::Bad; }

void *dev_type_malloc(size_t bytes, dev_type_t dev_type)
{
    if (reject(dev_type)) return NULL;
    void *mem = NULL;
    if (space_of(dev_type) == Space::Device) (void) crp_dev_malloc(&mem, bytes);
    else if (bytes > 0 && crp_host_malloc(&mem, bytes) != 0) mem = malloc(bytes);   // no GPU runtime: plain memory
    if (mem == NULL && bytes > 0) ERROR_PRINTF("Failed to malloc %zu bytes on device type %d\n", bytes, dev_type);
    return mem;
}

void dev_type_free(void *mem, dev_type_t dev_type)
{
    if (reject(dev_type) || mem == NULL) return;
    if (space_of(dev_type) == Space::Device) (void) crp_dev_free(mem);
    else if (crp_host_free(mem) != 0) free(mem);         // not a pinned allocation: it came from malloc
}

void dev_type_realloc(size_t *curr_bytes, size_t req_bytes, dev_type_t dev_type, void **mem)
{
    if (req_bytes <= *curr_bytes) return;                // grows only, contents are not kept
    dev_type_free(*mem, dev_type);
    *mem = dev_type_malloc(req_bytes, dev_type);
    *curr_bytes = (*mem != NULL) ? req_bytes : 0;
}

void dev_type_memset(void *mem, int value, size_t bytes, dev_type_t dev_type)
{
    if (reject(dev_type)) return;
    if (space_of(dev_type) == Space::Host)
    {
        memset(mem, value, bytes);
        return;
    }
    (void) crp_dev_memset(mem, value, bytes, NULL);
    (void) crp_stream_sync(NULL);
}

void dev_type_memcpy(void *dst, const void *src, size_t bytes, dev_type_t dst_dev_type, dev_type_t src_dev_type)
{
    const Space to = space_of(dst_dev_type), from = space_of(src_dev_type);
    if (to == Space::Bad || from == Space::Bad)
    {
        ERROR_PRINTF("Invalid dst device type %d or src device type %d\n", dst_dev_type, src_dev_type);
        return;
    }
    if (to == Space::Host && from == Space::Host)
    {
        memcpy(dst, src, bytes);
        return;
    }
    // crp_dev_memcpy kinds: 0 host -> device, 1 device -> host, 2 device -> device
    const int kind = from == Space::Host ? 0 : (to == Space::Host ? 1 : 2);
    (void) crp_dev_memcpy(dst, src, bytes, kind, NULL);
    (void) crp_stream_sync(NULL);
}

void dev_type_copy_matrix(size_t dt_size, const int nrow, const int ncol, const void *src, const int lds, void *dst,
                          const int ldd, dev_type_t dev_type)
{
    if (reject(dev_type)) return;
    if (space_of(dev_type) == Space::Host)
    {
        copy_matrix(dt_size, nrow, ncol, src, lds, dst, ldd, 1);
        return;
    }
    ASSERT_PRINTF(dt_size == 4 || dt_size == 8, "dt_size == 4 or 8 required for device memory\n");
    (void) crp_dev_memcpy2d(dst, dt_size * (size_t) ldd, src, dt_size * (size_t) lds, dt_size * (size_t) ncol, (size_t) nrow, 2,
                            NULL);
    (void) crp_stream_sync(NULL);
}

}  // extern "C"
