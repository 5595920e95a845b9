/* rccl_probe -- the device all-to-all of the MPI communicator back end (include/crp_mpi.h) on its own:
 * every rank sends a block of doubles to every rank (itself included) between device buffers and checks
 * what arrives.  Prints which transport was used ("rccl" needs one GPU per rank; ranks sharing a GPU
 * fall back to host staging).  mpiexec -np P examples/rccl_probe.exe */
#include <stdio.h>
#include <stdlib.h>
#include <mpi.h>
#include "crp_mpi.h"
#include "crpspmm_hip.h"

int main(int argc, char **argv)
{
    MPI_Init(&argc, &argv);
    int P, me, bad = 0;
    MPI_Comm_size(MPI_COMM_WORLD, &P);
    MPI_Comm_rank(MPI_COMM_WORLD, &me);
    crp_comm_t *comm = crp_mpi_comm_wrap(MPI_COMM_WORLD);
    const int rccl = crp_mpi_comm_uses_rccl(comm);
    const long long blk = 100000;                                /* doubles per (sender, receiver) pair, + sender rank */
    long long *cnt = (long long *) malloc(sizeof(long long) * P), *dsp = (long long *) malloc(sizeof(long long) * (P + 1));
    long long *rcnt = (long long *) malloc(sizeof(long long) * P), *rdsp = (long long *) malloc(sizeof(long long) * (P + 1));
    dsp[0] = rdsp[0] = 0;
    for (int q = 0; q < P; q++)
    {
        cnt[q] = blk + me;                                       /* what I send to q */
        rcnt[q] = blk + q;                                       /* what q sends to me */
        dsp[q + 1] = dsp[q] + cnt[q];
        rdsp[q + 1] = rdsp[q] + rcnt[q];
    }
    double *hs = (double *) malloc(sizeof(double) * dsp[P]), *hr = (double *) malloc(sizeof(double) * rdsp[P]);
    for (int q = 0; q < P; q++)
        for (long long i = 0; i < cnt[q]; i++) hs[dsp[q] + i] = 1000.0 * me + q + 1e-6 * (double) i;
    void *ds = NULL, *dr = NULL, *stream = NULL;
    if (crp_dev_malloc(&ds, sizeof(double) * dsp[P]) || crp_dev_malloc(&dr, sizeof(double) * rdsp[P]) || crp_stream_create(&stream)) bad = 1;
    if (!bad)
    {
        crp_dev_memcpy(ds, hs, sizeof(double) * dsp[P], 0, stream);
        crp_dev_memset(dr, 0, sizeof(double) * rdsp[P], stream);
        for (int rep = 0; rep < 3; rep++)
            comm->alltoallv_dev_f64(comm->ctx, (const double *) ds, cnt, dsp, (double *) dr, rcnt, rdsp, stream);
        crp_dev_memcpy(hr, dr, sizeof(double) * rdsp[P], 1, stream);
        crp_stream_sync(stream);
        for (int q = 0; q < P && !bad; q++)
            for (long long i = 0; i < rcnt[q]; i++)
                if (hr[rdsp[q] + i] != 1000.0 * q + me + 1e-6 * (double) i) { bad = 1; break; }
    }
    int any = 0;
    MPI_Allreduce(&bad, &any, 1, MPI_INT, MPI_MAX, MPI_COMM_WORLD);
    if (me == 0) printf("rccl_probe: %d ranks, transport %s, %s\n", P, rccl ? "rccl" : "host-staged", any ? "MISMATCH" : "ok");
    crp_dev_free(ds); crp_dev_free(dr); crp_stream_destroy(stream);
    comm->free(comm);
    free(cnt); free(dsp); free(rcnt); free(rdsp); free(hs); free(hr);
    MPI_Finalize();
    return any;
}
