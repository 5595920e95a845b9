// dev_type.cpp -- memory-space dispatch behind include/dev_type.h
// (behaviour of /root/reference/src/dev_type.c:13-150, HIP instead of CUDA).
#include <stdlib.h>
#include <string.h>
#include "crpspmm_hip.h"
#include "dev_type.h"

static inline bool on_device(dev_type_t t) { return t == DEV_TYPE_HIP || t == DEV_TYPE_HIP_RCCL; }

extern "C" {

int is_dev_type_valid(dev_type_t dev_type)
{
    return (dev_type == DEV_TYPE_HOST || dev_type == DEV_TYPE_HIP || dev_type == DEV_TYPE_HIP_RCCL) ? 1 : 0;
}

void *dev_type_malloc(size_t bytes, dev_type_t dev_type)
{
    void *mem = NULL;
    if (!is_dev_type_valid(dev_type))
    {
        ERROR_PRINTF("Invalid device type %d\n", dev_type);
        return mem;
    }
    if (dev_type == DEV_TYPE_HOST)
    {
        // pinned when a GPU runtime is usable, plain malloc otherwise (host-only tools)
        if (bytes > 0 && crp_host_malloc(&mem, bytes) != 0) mem = malloc(bytes);
    }
    else (void) crp_dev_malloc(&mem, bytes);
    if (bytes > 0 && mem == NULL) ERROR_PRINTF("Failed to malloc %zu bytes on device type %d\n", bytes, dev_type);
    return mem;
}

void dev_type_free(void *mem, dev_type_t dev_type)
{
    if (!is_dev_type_valid(dev_type))
    {
        ERROR_PRINTF("Invalid device type %d\n", dev_type);
        return;
    }
    if (mem == NULL) return;
    if (dev_type == DEV_TYPE_HOST)
    {
        int is_dev = 0;
        // pinned allocations are known to the runtime; plain malloc'd ones are not
        if (crp_host_free(mem) != 0) { (void) is_dev; free(mem); }
    }
    else (void) crp_dev_free(mem);
}

void dev_type_realloc(size_t *curr_bytes, size_t req_bytes, dev_type_t dev_type, void **mem)
{
    if (*curr_bytes >= req_bytes) return;
    dev_type_free(*mem, dev_type);
    *curr_bytes = 0;
    *mem = dev_type_malloc(req_bytes, dev_type);
    if (*mem != NULL) *curr_bytes = req_bytes;
}

void dev_type_memset(void *mem, int value, size_t bytes, dev_type_t dev_type)
{
    if (!is_dev_type_valid(dev_type))
    {
        ERROR_PRINTF("Invalid device type %d\n", dev_type);
        return;
    }
    if (dev_type == DEV_TYPE_HOST) memset(mem, value, bytes);
    else
    {
        (void) crp_dev_memset(mem, value, bytes, NULL);
        (void) crp_stream_sync(NULL);
    }
}

void dev_type_memcpy(void *dst, const void *src, size_t bytes, dev_type_t dst_dev_type, dev_type_t src_dev_type)
{
    if (!is_dev_type_valid(dst_dev_type) || !is_dev_type_valid(src_dev_type))
    {
        ERROR_PRINTF("Invalid dst device type %d or src device type %d\n", dst_dev_type, src_dev_type);
        return;
    }
    const bool d = on_device(dst_dev_type), s = on_device(src_dev_type);
    if (!d && !s)
    {
        memcpy(dst, src, bytes);
        return;
    }
    const int kind = (d && !s) ? 0 : (!d && s) ? 1 : 2;
    (void) crp_dev_memcpy(dst, src, bytes, kind, NULL);
    (void) crp_stream_sync(NULL);
}

void dev_type_copy_matrix(size_t dt_size, const int nrow, const int ncol, const void *src, const int lds, void *dst,
                          const int ldd, dev_type_t dev_type)
{
    if (!is_dev_type_valid(dev_type))
    {
        ERROR_PRINTF("Invalid device type %d\n", dev_type);
        return;
    }
    if (dev_type == DEV_TYPE_HOST)
    {
        copy_matrix(dt_size, nrow, ncol, src, lds, dst, ldd, 1);
        return;
    }
    ASSERT_PRINTF(dt_size == 4 || dt_size == 8, "dt_size == 4 or 8 required for device memory\n");
    (void) crp_dev_memcpy2d(dst, dt_size * (size_t) ldd, src, dt_size * (size_t) lds, dt_size * (size_t) ncol,
                            (size_t) nrow, 2, NULL);
    (void) crp_stream_sync(NULL);
}

}  // extern "C"
