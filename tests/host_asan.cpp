// Host-side format builders, planner and ingest under AddressSanitizer / UBSan (CPU only; built and
// run by tests/test_host_sanitizers.py).  No device code is linked: everything here is the host
// half of the product (csrc/panel_format.cpp, spmat_part.cpp, mmio_utils.cpp, host_support.cpp).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "dev_type.h"
#include "locality.h"
#include "panel_format.h"
#include "mmio_utils.h"
#include "spmat_part.h"
#include "utils.h"

extern "C" int crp_csr_cache_write(const char *, int, int, const int *, const int *, const double *);
extern "C" int crp_csr_cache_read(const char *, int *, int *, int **, int **, double **);

// host_support.cpp also holds the device branch of dev_type_*: with no GPU library linked, the
// device ABI it calls is stubbed to "no runtime" (the host branches then take plain malloc)
extern "C" {
int crp_dev_malloc(void **p, size_t) { *p = NULL; return -1; }
int crp_dev_free(void *) { return -1; }
int crp_dev_memset(void *, int, size_t, void *) { return -1; }
int crp_dev_memcpy(void *, const void *, size_t, int, void *) { return -1; }
int crp_dev_memcpy2d(void *, size_t, const void *, size_t, size_t, size_t, int, void *) { return -1; }
int crp_host_malloc(void **p, size_t) { *p = NULL; return -1; }
int crp_host_free(void *) { return -1; }
int crp_stream_sync(void *) { return -1; }
}

static uint64_t rng_state = 88172645463325252ull;
static uint32_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t) (rng_state >> 11);
}

struct Csr { int m, k; std::vector<int> rp, ci; std::vector<double> va; };

static Csr banded(int m, const std::vector<int> &offs)
{
    Csr a; a.m = a.k = m; a.rp.assign(m + 1, 0);
    std::vector<int> all;
    for (int d : offs) { all.push_back(d); all.push_back(-d); }
    all.push_back(0);
    std::sort(all.begin(), all.end());
    for (int i = 0; i < m; i++)
    {
        for (int d : all)
            if (i + d >= 0 && i + d < m) { a.ci.push_back(i + d); a.va.push_back(1.0 + (rnd() % 100) * 0.01); }
        a.rp[i + 1] = (int) a.ci.size();
    }
    return a;
}

static Csr random_csr(int m, int k, int maxdeg, bool dup, bool two_source)
{
    Csr a; a.m = m; a.k = k; a.rp.assign(m + 1, 0);
    for (int i = 0; i < m; i++)
    {
        const int deg = (i % 7 == 3) ? 0 : (int) (rnd() % (maxdeg + 1));
        std::vector<int> c;
        for (int t = 0; t < deg; t++) c.push_back((int) (rnd() % k));
        std::sort(c.begin(), c.end());
        if (!dup) c.erase(std::unique(c.begin(), c.end()), c.end());
        for (int x : c)
        {
            a.ci.push_back(two_source && x >= k / 2 ? ~(x - k / 2) : x);
            a.va.push_back((rnd() % 2000) * 0.001 - 1.0);
        }
        a.rp[i + 1] = (int) a.ci.size();
    }
    return a;
}

// every CSR nonzero must be reachable through the format exactly once
static void check_format(const Csr &a, const crp::PanelHost &h)
{
    const int R = h.R;
    for (size_t p = 0; p < h.pmap.size(); p++)
    {
        const uint32_t slot = h.pmap[p];
        const size_t q = slot / R, r = slot % R;
        if (q >= h.pcol.size() || h.pval[q * R + r] != a.va[p] || h.pcol[q] != a.ci[p]) { printf("FAIL pmap %zu\n", p); exit(1); }
        if (!((h.pmask4[q >> 2] >> (8 * (q & 3))) >> r & 1)) { printf("FAIL mask %zu\n", p); exit(1); }
    }
    std::vector<int> cnt((size_t) h.npanel, 0);
    for (int x : h.porder)
        if (x >= 0) { if (x >= h.npanel) { printf("FAIL order\n"); exit(1); } cnt[(size_t) x]++; }
    for (int c : cnt) if (c != 1) { printf("FAIL order is not a cover\n"); exit(1); }
}

int main()
{
    std::vector<Csr> mats;
    mats.push_back(banded(300 * 8 * 5 + 13, {1, 2, 3, 300, 301, 2400, 2401}));       // stride lattice, ragged end
    mats.push_back(banded(5000, {1, 2, 3, 40, 900}));
    mats.push_back(random_csr(777, 1234, 40, true, false));
    mats.push_back(random_csr(301, 500, 9, false, true));
    mats.push_back(random_csr(5, 9, 3, false, false));
    mats.push_back(random_csr(0, 4, 3, false, false));
    const char *orders[] = {NULL, "0", "1", "2", "3"};
    for (const Csr &a : mats)
        for (const char *o : orders)
        {
            if (o) setenv("CRPSPMM_PANEL_ORDER", o, 1); else unsetenv("CRPSPMM_PANEL_ORDER");
            for (int R : {4, 8})
            {
                crp::PanelHost h;
                crp::build_panels(a.m, a.rp.data(), a.ci.data(), a.va.data(), R, &h);
                check_format(a, h);
                (void) crp::count_panel_entries(a.m, a.rp.data(), a.ci.data(), R);
            }
            crp::PanelHost h8;
            crp::build_panels(a.m, a.rp.data(), a.ci.data(), a.va.data(), 8, &h8, false);
            crp::TeamHost t;
            crp::build_teams(h8, a.m, a.rp.data(), a.ci.data(), &t);
            long long own = 0;
            for (uint32_t m : t.tmask) for (int w = 0; w < 4; w++) own += ((m >> (8 * w)) & 0xFF) != 0;
            if (own != t.tvoff.back()) { printf("FAIL team value streams %lld %lld\n", own, t.tvoff.back()); return 1; }
        }
    unsetenv("CRPSPMM_PANEL_ORDER");

    // team2 streams (variant 5): clustered / lattice / consecutive teams, super-teams, launch grid; locality order
    {
        std::vector<Csr> t2 = mats;
        t2.push_back(banded(9120, {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 304, 305, 306, 307, 308, 309, 3040, 3041, 3042, 3043, 3044, 3045}));
        t2.push_back(banded(300 * 40 + 5, {1, 2, 300, 301, 302}));          // enough teams for super-teams (>= 128)
        t2.push_back(banded(96 * 32 * 32 + 3, {1, 2, 3, 96, 97, 96 * 32, 96 * 32 + 1}));   // a lattice with >= 1024 teams: the order search (team_order.cpp)
        for (const Csr &a : t2)
        {
            crp::PanelHost h8;
            crp::build_panels(a.m, a.rp.data(), a.ci.data(), a.va.data(), 8, &h8, false);
            std::vector<int> perm, pos;
            bool square = a.m == a.k;
            for (int c : a.ci) square = square && c >= 0;
            if (square && crp::locality_reorder(a.m, a.k, a.rp.data(), a.ci.data(), 8, &perm))
            {
                if ((int) perm.size() != a.m) { printf("FAIL locality size\n"); return 1; }
                pos.assign((size_t) a.m, -1);
                for (int i = 0; i < a.m; i++) pos[(size_t) perm[(size_t) i]] = i;
                for (int v : pos) if (v < 0) { printf("FAIL locality is not a permutation\n"); return 1; }
            }
            crp::Team2Host t;
            crp::build_team2(h8, a.m, a.rp.data(), a.ci.data(), &t, pos.empty() ? nullptr : pos.data());
            std::vector<int> seen((size_t) std::max(t.nteam, 1), 0);
            if (t.tgrid.size() % 8) { printf("FAIL tgrid size\n"); return 1; }
            for (int g : t.tgrid)
                if (g >= 0) { if (g >= t.nteam) { printf("FAIL tgrid entry\n"); return 1; } seen[(size_t) g]++; }
            for (int g = 0; g < t.nteam; g++) if (seen[(size_t) g] != 1) { printf("FAIL tgrid is not a cover\n"); return 1; }
            long long parts = 0;
            for (int g = 0; g < t.nteam; g++) parts += t.tinfo[(size_t) g * 4 + 2];
            if (parts != t.parts || t.tvoff.back() * crp::TEAM2_VUNIT != t.nvalues || (long long) t.tval.size() != t.nvalues)
            { printf("FAIL team2 parts / values %lld %lld %lld %lld\n", parts, t.parts, t.tvoff.back(), t.nvalues); return 1; }
            if (t.vmap.size() != a.ci.size()) { printf("FAIL team2 vmap\n"); return 1; }
            for (size_t p2 = 0; p2 < t.vmap.size(); p2++)
                if (t.vmap[p2] >= t.tval.size() || t.tval[t.vmap[p2]] != a.va[p2]) { printf("FAIL team2 vmap entry %zu\n", p2); return 1; }
            // the streams of the row-owner team kernel (variant 7), both lane groupings
            for (int G : {4, 2})
            {
                crp::Team2RHost tr;
                tr.G = G;
                if (!crp::build_team2r(h8, a.m, a.rp.data(), a.ci.data(), &tr, pos.empty() ? nullptr : pos.data())) { printf("FAIL team2r refused\n"); return 1; }
                if (tr.tgrid.size() % 8 || tr.tent.size() != tr.tgrid.size() * 256 || (long long) tr.tval.size() < tr.nwords || tr.vmap.size() != a.ci.size())
                { printf("FAIL team2r sizes\n"); return 1; }
                for (size_t p2 = 0; p2 < tr.vmap.size(); p2++)
                    if ((long long) tr.vmap[p2] >= tr.nwords || tr.tval[tr.vmap[p2]] != a.va[p2]) { printf("FAIL team2r vmap entry %zu\n", p2); return 1; }
                long long rounds = 0;
                for (int g = 0; g < tr.nteam; g++) rounds += tr.tinfo[(size_t) g * 2];
                if (rounds != tr.rounds) { printf("FAIL team2r rounds\n"); return 1; }
            }
            // what the device path does (hip_api.hip, panel_skeleton): panels without values, the teams of the first format seed the
            // next ones, values scattered through vmap -- the same streams as the full builds above
            {
                auto differs = [](const void *x, const void *y, size_t bytes) { return bytes > 0 && memcmp(x, y, bytes) != 0; };
                crp::PanelHost sk;
                crp::build_panels(a.m, a.rp.data(), a.ci.data(), nullptr, 8, &sk, false, false);
                if (!sk.pval.empty() || sk.pcol.size() != h8.pcol.size() || differs(sk.pmap.data(), h8.pmap.data(), sizeof(uint32_t) * sk.pmap.size()))
                { printf("FAIL structure-only panels\n"); return 1; }
                crp::TeamSeed seed;
                crp::Team2Host t0;
                t0.compact = t.compact;
                crp::build_team2(sk, a.m, a.rp.data(), a.ci.data(), &t0, pos.empty() ? nullptr : pos.data(), &seed);
                if (!t0.tval.empty() || t0.nvalues != t.nvalues || t0.trec.size() != t.trec.size() || t0.tgrid != t.tgrid || t0.tinfo != t.tinfo || t0.tvoff != t.tvoff
                    || t0.vmap != t.vmap || t0.tpro != t.tpro || differs(t0.trec.data(), t.trec.data(), sizeof(uint32_t) * t.trec.size()))
                { printf("FAIL team2 from structure-only panels\n"); return 1; }
                if (a.m >= 16 && !seed.valid) { printf("FAIL seed not filled\n"); return 1; }
                for (int G : {4, 2})
                {
                    crp::Team2RHost full, tr;
                    full.G = tr.G = G;
                    if (!crp::build_team2r(h8, a.m, a.rp.data(), a.ci.data(), &full, pos.empty() ? nullptr : pos.data())) { printf("FAIL team2r refused\n"); return 1; }
                    if (!crp::build_team2r(sk, a.m, a.rp.data(), a.ci.data(), &tr, pos.empty() ? nullptr : pos.data(), &seed)) { printf("FAIL team2r refused (seeded)\n"); return 1; }
                    if (tr.tgrid != full.tgrid || tr.tinfo != full.tinfo || tr.tvoff != full.tvoff || tr.vmap != full.vmap || tr.tval.size() != full.tval.size()
                        || tr.trec.size() != full.trec.size() || differs(tr.trec.data(), full.trec.data(), sizeof(uint32_t) * full.trec.size())
                        || tr.tent.size() != full.tent.size() || differs(tr.tent.data(), full.tent.data(), sizeof(uint32_t) * full.tent.size()))
                    { printf("FAIL seeded team2r structure\n"); return 1; }
                    for (size_t p2 = 0; p2 < tr.vmap.size(); p2++) tr.tval[tr.vmap[p2]] = a.va[p2];
                    if (differs(tr.tval.data(), full.tval.data(), sizeof(double) * full.tval.size())) { printf("FAIL seeded team2r streams\n"); return 1; }
                }
            }
        }
    }

    // planner
    {
        const Csr a = banded(6000, {1, 2, 3, 40, 41, 900});
        for (int P : {1, 2, 3, 4, 6, 8, 12})
        {
            std::vector<int> rb(P + 1), sizes(P);
            csr_mat_row_partition(a.m, a.rp.data(), P, rb.data());
            int tot = 0;
            csr_mat_row_part_comm_size(a.m, a.k, a.rp.data(), a.ci.data(), P, rb.data(), rb.data(), sizes.data(), &tot);
            int pm, pn, *a0, *br, *ac, *bc;
            size_t cost;
            calc_spmm_part2d_from_1d(P, a.m, 48, a.k, rb.data(), a.rp.data(), a.ci.data(), 1, &pm, &pn, &cost, &a0, &br, &ac, &bc, 0);
            if (pm * pn != P || a0[P] != a.m || ac[pm] != a.m || bc[pn] != 48) { printf("FAIL planner\n"); return 1; }
            free(a0); free(br); free(ac); free(bc);
            int *fac = NULL;
            const int nf = prime_factorization(P, &fac);
            int prod = 1;
            for (int i = 0; i < nf; i++) prod *= fac[i];
            free(fac);
            if (prod != P) { printf("FAIL factors\n"); return 1; }
        }
    }

    // ingest: parallel path (> 200k entries), cache round trip
    {
        const Csr a = banded(40000, {1, 2, 3, 50, 700});
        const char *fn = "/tmp/crp_asan_big.mtx";
        FILE *f = fopen(fn, "w");
        long cnt = 0;
        for (int i = 0; i < a.m; i++) for (int p = a.rp[i]; p < a.rp[i + 1]; p++) cnt += a.ci[p] <= i;
        fprintf(f, "%%%%MatrixMarket matrix coordinate real symmetric\n%% c\n%d %d %ld\n", a.m, a.m, cnt);
        for (int i = 0; i < a.m; i++)
            for (int p = a.rp[i]; p < a.rp[i + 1]; p++)
                if (a.ci[p] <= i) fprintf(f, "%d %d %.17g\n", i + 1, a.ci[p] + 1, a.va[p]);
        fclose(f);
        int nr, nc, nnz, *row, *col, *rp, *ci;
        double *val, *va;
        if (mm_read_sparse_RPI(fn, 0, &nr, &nc, &nnz, &row, &col, &val) != 0 || nnz != (int) a.ci.size()) { printf("FAIL ingest\n"); return 1; }
        coo2csr(nr, nc, nnz, row, col, val, &rp, &ci, &va);
        if (memcmp(rp, a.rp.data(), sizeof(int) * (a.m + 1)) || memcmp(ci, a.ci.data(), sizeof(int) * a.ci.size())) { printf("FAIL coo2csr\n"); return 1; }
        const char *cf = "/tmp/crp_asan_big.crpcsr";
        if (crp_csr_cache_write(cf, nr, nc, rp, ci, va) != 0) { printf("FAIL cache write\n"); return 1; }
        int m2, k2, *rp2, *ci2;
        double *va2;
        if (crp_csr_cache_read(cf, &m2, &k2, &rp2, &ci2, &va2) != 0 || m2 != nr || memcmp(va2, va, sizeof(double) * (size_t) nnz)) { printf("FAIL cache read\n"); return 1; }
        free(row); free(col); free(val); free(rp); free(ci); free(va); free(rp2); free(ci2); free(va2);
        remove(fn);
        remove(cf);
    }
    // utils.h / dev_type.h, host branches
    {
        for (int len : {0, 1, 7, 100})
            for (int nblk : {1, 3, 8})
            {
                int pos, size, sum = 0;
                for (int b = 0; b < nblk; b++)
                {
                    calc_block_spos_size(len, nblk, b, &pos, &size);
                    if (pos != sum) { printf("FAIL block split\n"); return 1; }
                    sum += size;
                }
                calc_block_spos_size(len, nblk, nblk, &pos, &size);
                if (sum != len || pos != len) { printf("FAIL block split end\n"); return 1; }     /* (size of the sentinel: len / nblk, as the reference) */
            }
        size_t cap = 0;
        void *buf = NULL;
        dev_type_realloc(&cap, 1000, DEV_TYPE_HOST, &buf);
        dev_type_memset(buf, 7, 1000, DEV_TYPE_HOST);
        std::vector<char> dst(1000);
        dev_type_memcpy(dst.data(), buf, 1000, DEV_TYPE_HOST, DEV_TYPE_HOST);
        dev_type_copy_matrix(4, 10, 5, buf, 20, dst.data(), 7, DEV_TYPE_HOST);
        dev_type_free(buf, DEV_TYPE_HOST);
        if (dst[999] != 7 || is_dev_type_valid((dev_type_t) 9) || dev_type_malloc(8, (dev_type_t) 9) != NULL) { printf("FAIL dev_type\n"); return 1; }
        double x[3] = {3, 4, 0}, y[3] = {3, 4, 2}, nx, ne;
        calc_err_2norm(3, x, y, &nx, &ne);
        if (calc_2norm(3, x) != 5.0 || nx != 5.0 || ne != 2.0) { printf("FAIL norms\n"); return 1; }
    }
    printf("HOST_ASAN_OK\n");
    return 0;
}
