/*
 * dev_type.h -- memory-space dispatch used by the redistribution engine; same entry
 * points and enum values as /root/reference/src/dev_type.h:9-57, re-targeted from CUDA
 * to HIP (gfx950):
 *   0 DEV_TYPE_HOST                                   host memory
 *   1 DEV_TYPE_HIP       (reference: DEV_TYPE_CUDA)   hipMalloc memory, payloads staged through the host
 *   2 DEV_TYPE_HIP_RCCL  (reference: ..._MPI_DIRECT)  hipMalloc memory, payloads exchanged device to device
 * The reference's enumerator names are kept as aliases so that callers compile unchanged.
 * Every function validates dev_type and prints "[ERROR] ... Invalid device type" like the
 * reference (src/dev_type.c:13-150); there is no silent fallback between spaces.
 */
#ifndef CRP_DEV_TYPE_H
#define CRP_DEV_TYPE_H

#include <stddef.h>
#include <stdint.h>
#include "utils.h"

typedef enum
{
    DEV_TYPE_HOST = 0,
    DEV_TYPE_HIP = 1,
    DEV_TYPE_HIP_RCCL = 2,
    DEV_TYPE_CUDA = DEV_TYPE_HIP,
    DEV_TYPE_CUDA_MPI_DIRECT = DEV_TYPE_HIP_RCCL
} dev_type_t;

#ifdef __cplusplus
extern "C" {
#endif

int   is_dev_type_valid(dev_type_t dev_type);
/* host memory is pinned (hipHostMalloc) so that it can mirror device buffers, as the
 * reference does under USE_CUDA (src/dev_type.c:41-44) */
void *dev_type_malloc(size_t bytes, dev_type_t dev_type);
void  dev_type_free(void *mem, dev_type_t dev_type);
/* grow-only: reallocates (contents dropped) when req_bytes > *curr_bytes (src/dev_type.c:78-85) */
void  dev_type_realloc(size_t *curr_bytes, size_t req_bytes, dev_type_t dev_type, void **mem);
void  dev_type_memset(void *mem, int value, size_t bytes, dev_type_t dev_type);
void  dev_type_memcpy(void *dst, const void *src, size_t bytes, dev_type_t dst_dev_type, dev_type_t src_dev_type);
/* row-major rectangle copy inside one memory space; device spaces need dt_size 4 or 8 */
void  dev_type_copy_matrix(size_t dt_size, const int nrow, const int ncol, const void *src, const int lds,
                           void *dst, const int ldd, dev_type_t dev_type);

#ifdef __cplusplus
}
#endif
#endif
