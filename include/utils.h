/*
 * utils.h -- helper routines of the CRP-SpMM library, same names, argument
 * meaning and behaviour as /root/reference/src/utils.h:104-191 (callers: the
 * planner, the engines and the example drivers).
 */
#ifndef CRP_UTILS_H
#define CRP_UTILS_H

#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#ifdef __cplusplus
#include <cassert>
extern "C" {
#else
#include <assert.h>
#endif

#define INT_MSIZE sizeof(int)
#define DBL_MSIZE sizeof(double)
#ifndef MIN
#define MIN(a, b) ((a) < (b) ? (a) : (b))
#endif
#ifndef MAX
#define MAX(a, b) ((a) > (b) ? (a) : (b))
#endif

/* Message macros: same prefixes and streams as src/utils.h:26-68. */
#define CRP_MSG_(stream, tag, fmt, ...)                                              \
    do {                                                                             \
        fprintf(stream, "[" tag "] %s, %d: " fmt, __FILE__, __LINE__, ##__VA_ARGS__); \
        fflush(stream);                                                              \
    } while (0)
#define INFO_PRINTF(fmt, ...)    CRP_MSG_(stdout, "INFO", fmt, ##__VA_ARGS__)
#define DEBUG_PRINTF(fmt, ...)   CRP_MSG_(stderr, "DEBUG", fmt, ##__VA_ARGS__)
#define WARNING_PRINTF(fmt, ...) CRP_MSG_(stderr, "WARNING", fmt, ##__VA_ARGS__)
#define ERROR_PRINTF(fmt, ...)   CRP_MSG_(stderr, "ERROR", fmt, ##__VA_ARGS__)
#define ASSERT_PRINTF(expr, fmt, ...)                    \
    do {                                                 \
        if (!(expr)) {                                   \
            CRP_MSG_(stderr, "FATAL", fmt, ##__VA_ARGS__); \
            assert(expr);                                \
            abort();                                     \
        }                                                \
    } while (0)

/* Integer knob from the environment (src/utils.h:71-87): out-of-range values
 * fall back to the default; an override is announced with the reference's
 * "[INFO] ... Overriding parameter ..." line when print_info is nonzero. */
#define GET_ENV_INT_VAR(var, env_str, var_str, default_val, min_val, max_val, print_info)      \
    do {                                                                                       \
        const char *crp_env_p_ = getenv(env_str);                                              \
        var = default_val;                                                                     \
        if (crp_env_p_ != NULL) {                                                              \
            var = atoi(crp_env_p_);                                                            \
            if (var < (min_val) || var > (max_val)) var = default_val;                         \
            if ((print_info) && var != (default_val))                                          \
                INFO_PRINTF("Overriding parameter %s: %d (default) --> %d (runtime)\n",        \
                            var_str, default_val, var);                                        \
        }                                                                                      \
    } while (0)

/* Wall-clock seconds (gettimeofday resolution, src/utils.c:15-22). */
double get_wtime_sec(void);
/* Even split of len into nblk blocks, remainder to the first blocks; iblk in
 * [0, nblk], iblk == nblk gives (len, 0); invalid iblk gives (-1, 0)
 * (src/utils.c:26-48). */
void calc_block_spos_size(const int len, const int nblk, const int iblk, int *blk_spos, int *blk_size);
void *malloc_aligned(size_t size, size_t alignment);
void free_aligned(void *mem);
/* Naive-sum 2-norms (src/utils.c:66-89). */
double calc_2norm(const int len, const double *x);
void calc_err_2norm(const int len, const double *x0, const double *x1, double *x0_2norm_, double *err_2norm_);
/* Row-major rectangle copy, dt_size bytes per element (src/utils.c:92-119). */
void copy_matrix(const size_t dt_size, const int nrow, const int ncol, const void *src, const int lds,
                 void *dst, const int ldd, const int use_omp);
/* dtype 0 int / 1 double; stype 0 row-major / 1 column-major (src/utils.c:122-156). */
void print_matrix(const int dtype, const int stype, const void *mat, const int ldm, const int nrow,
                  const int ncol, const char *fmt, const char *name);
void dump_binary(const char *fname, void *data, const size_t bytes);

#ifdef __cplusplus
}
#endif
#endif
