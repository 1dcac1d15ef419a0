// Host-only build of the library's plain-C++ parts under AddressSanitizer + UBSan (never the GPU build):
//   kmer_output_loader.inc (output.bin -> CSR), projection_tables.inc (the embed kernel's lookup tables),
//   csr_compact.inc (dead-feature filter), knn_plan.inc (launch planner), overlaps_writer.inc (overlaps.tsv).
// tests/test_host_san.py builds this with g++ -fsanitize=address,undefined and drives it; each command
// prints a result line that the test compares with what libfedrann_hip.so returns for the same input.
//
//   host_san loader PATH N_FEATURES THREADS       -> "rc=<code> R=.. nnz=.. sums=<4 weighted sums>" | "rc=<code> err=<msg>"
//   host_san loader-stale PATH N_FEATURES         -> load with capacities that no longer match: must fail cleanly
//   host_san tables SEED N_FEATURES D             -> builds tables for a random very-sparse P, checks them, compacts a CSR
//   host_san floats N  (hex float32 words on stdin) -> the writer's text of each
//   host_san overlaps OUT THREADS                 -> writes a random neighbour graph (whole, and as two appended blocks)
//   host_san plan-print NQ NT D K SHAPE           -> one plan (devtools)
//   host_san plan                                 -> sweeps the planner over edge sizes, checks invariants
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <random>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/fedrann_hip.h"
#include "../../fedrann_amd/csrc/host_common.inc"
#include "../../fedrann_amd/csrc/knn_plan.inc"
#include "../../fedrann_amd/csrc/projection_tables.inc"
#include "../../fedrann_amd/csrc/csr_compact.inc"
#include "../../fedrann_amd/csrc/kmer_output_loader.inc"
#include "../../fedrann_amd/csrc/reads_parser.inc"
#include "../../fedrann_amd/csrc/overlaps_writer.inc"

template <typename T>
static uint64_t wsum(const std::vector<T> &v) {  // sum of (i + 1) * v[i] mod 2^64 (numpy can restate it)
    uint64_t s = 0;
    for (size_t i = 0; i < v.size(); ++i)
        s += (uint64_t)(i + 1) * (uint64_t)(typename std::make_unsigned<T>::type)v[i];
    return s;
}

static int cmd_loader(const char *path, long long F, int threads, bool stale) {
    int64_t R = 0, nnz = 0, nb = 0;
    int rc = fdr_kmer_output_scan(path, &R, &nnz, &nb);
    if (rc) {
        printf("rc=%d err=%s\n", rc, g_err);
        return 0;
    }
    std::vector<int64_t> indptr((size_t)(2 * R + 1)), name_off((size_t)(R + 1));
    std::vector<int32_t> indices((size_t)(2 * nnz));
    std::vector<char> names((size_t)nb);
    rc = fdr_kmer_output_load(path, F, threads, stale ? R + 1 : R, nnz, nb, indptr.data(), indices.data(),
                              name_off.data(), names.data());
    if (rc) {
        printf("rc=%d err=%s\n", rc, g_err);
        return 0;
    }
    printf("rc=0 R=%lld nnz=%lld sums=%llu,%llu,%llu,%llu\n", (long long)R, (long long)nnz,
           (unsigned long long)wsum(indptr), (unsigned long long)wsum(indices), (unsigned long long)wsum(name_off),
           (unsigned long long)wsum(names));
    return 0;
}

// records [lo, hi) through the ranged loader, arrays sized EXACTLY (an overrun is an ASan report)
static int cmd_loader_range(const char *path, long long F, int threads, long long lo, long long hi, int with_names) {
    int64_t R = 0, nnz = 0, nb = 0;
    int rc = fdr_kmer_output_scan_range(path, lo, hi, &R, &nnz, &nb);
    if (rc) {
        printf("rc=%d err=%s\n", rc, g_err);
        return 0;
    }
    std::vector<int64_t> indptr((size_t)(2 * (hi - lo) + 1)), name_off(with_names ? (size_t)(R + 1) : 0);
    std::vector<int32_t> indices((size_t)(2 * nnz));
    std::vector<char> names(with_names ? (size_t)nb : 0);
    rc = fdr_kmer_output_load_range(path, F, threads, R, lo, hi, nnz, nb, indptr.data(), indices.data(),
                                    with_names ? name_off.data() : nullptr, with_names ? names.data() : nullptr);
    if (rc) {
        printf("rc=%d err=%s\n", rc, g_err);
        return 0;
    }
    printf("rc=0 R=%lld nnz=%lld sums=%llu,%llu,%llu,%llu\n", (long long)R, (long long)nnz,
           (unsigned long long)wsum(indptr), (unsigned long long)wsum(indices), (unsigned long long)wsum(name_off),
           (unsigned long long)wsum(names));
    return 0;
}

// a FASTA / FASTQ file streamed in pieces of `chunk` bytes through fdr_reads_scan / fdr_reads_parse, every array
// sized EXACTLY (an overrun is an ASan report); the records go back out through fdr_kmer_output_append with one
// "index" per record (its sequence length), which exercises the writer under the sanitizers as well
static int cmd_reads(const char *path, long long chunk, int ids_as_fasta, const char *out_path) {
    FILE *f = fopen(path, "rb");
    if (!f) return 1;
    std::vector<uint8_t> buf;
    size_t have = 0;
    bool eof = false;
    int is_fastq = -1;
    long long n_rec = 0, n_bases = 0;
    unsigned long long s_seq = 0, s_ids = 0;
    {
        FILE *o = fopen(out_path, "wb");
        if (!o) return 1;
        const unsigned char hdr[16] = {'K', 'M', 'E', 'R', 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        fwrite(hdr, 1, 16, o);
        fclose(o);
    }
    while (!eof) {
        std::vector<uint8_t> piece(have + (size_t)chunk);  // (a fresh, exactly sized buffer per piece)
        if (have) memcpy(piece.data(), buf.data(), have);
        const size_t got = fread(piece.data() + have, 1, (size_t)chunk, f);
        piece.resize(have + got);
        eof = got < (size_t)chunk;
        if (is_fastq < 0) is_fastq = !piece.empty() && piece[0] == '@';
        int64_t used = 0, R = 0, nb = 0;
        int rc = fdr_reads_scan(piece.data(), (int64_t)piece.size(), is_fastq, ids_as_fasta, eof ? 1 : 0, &used, &R, &nb);
        if (rc) {
            printf("rc=%d err=%s\n", rc, g_err);
            return 0;
        }
        std::vector<uint8_t> seqs((size_t)nb);
        std::vector<int64_t> off((size_t)R + 1), span((size_t)(2 * R));
        rc = fdr_reads_parse(piece.data(), used, is_fastq, ids_as_fasta, R, nb, seqs.data(), off.data(), span.data());
        if (rc) {
            printf("rc=%d err=%s\n", rc, g_err);
            return 0;
        }
        std::vector<int64_t> name_off((size_t)R + 1, 0), indptr((size_t)R + 1, 0);
        std::vector<char> names;
        std::vector<int32_t> idx((size_t)R);
        for (int64_t r = 0; r < R; ++r) {
            for (int64_t i = span[(size_t)(2 * r)]; i < span[(size_t)(2 * r + 1)]; ++i) {
                s_ids = s_ids * 1099511628211ull + piece[(size_t)i];
                names.push_back((char)(piece[(size_t)i] < 32 || piece[(size_t)i] > 126 ? '?' : piece[(size_t)i]));
            }
            s_ids = s_ids * 1099511628211ull + 255;
            name_off[(size_t)r + 1] = (int64_t)names.size();
            idx[(size_t)r] = (int32_t)(off[(size_t)r + 1] - off[(size_t)r]);
            indptr[(size_t)r + 1] = r + 1;
        }
        for (uint8_t c : seqs) s_seq = s_seq * 1099511628211ull + c;
        rc = fdr_kmer_output_append(out_path, R, name_off.data(), names.data(), indptr.data(), idx.data());
        if (rc) {
            printf("rc=%d err=%s\n", rc, g_err);
            return 0;
        }
        n_rec += R;
        n_bases += nb;
        buf.assign(piece.begin() + used, piece.end());
        have = buf.size();
    }
    fclose(f);
    printf("rc=0 R=%lld bases=%lld seq=%llu ids=%llu\n", n_rec, n_bases, s_seq, s_ids);
    return 0;
}

static int cmd_tables(unsigned seed, long long F, int d) {
    std::mt19937_64 rng(seed);
    // a very sparse P: each feature row is non-empty with probability ~ d / sqrt(F) (capped), 1-3 entries
    std::vector<int64_t> indptr((size_t)F + 1, 0);
    std::vector<int32_t> cols;
    std::vector<float> vals;
    const double p = std::min(0.5, (double)d / std::sqrt((double)F));
    for (long long f = 0; f < F; ++f) {
        if ((rng() >> 11) * (1.0 / 9007199254740992.0) < p) {
            const int n = 1 + (int)(rng() % 3);
            for (int i = 0; i < n; ++i) {
                cols.push_back((int32_t)(rng() % (unsigned)d));
                vals.push_back((float)((int)(rng() % 2001) - 1000) / 256.0f);
            }
        }
        indptr[(size_t)f + 1] = (int64_t)cols.size();
    }
    ProjectionTables T;
    int rc = build_projection_tables(F, d, indptr.data(), cols.data(), vals.data(), T);
    if (rc) {
        printf("rc=%d err=%s\n", rc, g_err);
        return 1;
    }
    // invariants: bit <=> non-empty row, prefix counts, rowinfo mirrors the CSR
    unsigned rows = 0;
    for (long long f = 0; f < F; ++f) {
        const bool bit = (T.ftab[(size_t)(f >> 5)].x >> (f & 31)) & 1u;
        const bool nonempty = indptr[(size_t)f + 1] > indptr[(size_t)f];
        if (bit != nonempty) return printf("FAIL bit %lld\n", f), 1;
        if ((f & 31) == 0 && T.ftab[(size_t)(f >> 5)].y != rows) return printf("FAIL prefix %lld\n", f), 1;
        if (nonempty) {
            const PU4 &ri = T.rowinfo[rows++];
            if (ri.x != (unsigned)indptr[(size_t)f] || ri.y != (unsigned)(indptr[(size_t)f + 1] - indptr[(size_t)f]) ||
                ri.z != (unsigned)cols[(size_t)indptr[(size_t)f]])
                return printf("FAIL rowinfo %lld\n", f), 1;
        }
    }
    if (rows != T.rows) return printf("FAIL rows\n"), 1;
    // bad inputs must be refused, not read out of bounds
    {
        std::vector<int64_t> bad = indptr;
        bad[(size_t)F / 2] = bad[(size_t)F] + 5;
        ProjectionTables U;
        if (build_projection_tables(F, d, bad.data(), cols.data(), vals.data(), U) == FDR_OK) return printf("FAIL monotone\n"), 1;
        if (!cols.empty()) {
            std::vector<int32_t> bc = cols;
            bc[bc.size() / 2] = d;
            if (build_projection_tables(F, d, indptr.data(), bc.data(), vals.data(), U) == FDR_OK) return printf("FAIL column\n"), 1;
        }
        if (build_projection_tables(F, FDR_MAX_DIM + 1, indptr.data(), cols.data(), vals.data(), U) == FDR_OK) return printf("FAIL dim\n"), 1;
    }
    // compaction of a random CSR (incl. empty rows, ids outside [0, F)) at 1 and 5 threads == a serial filter
    std::vector<uint32_t> bits(T.ftab.size());
    for (size_t w = 0; w < bits.size(); ++w) bits[w] = T.ftab[w].x;
    const int64_t n_rows = 20000;
    std::vector<int64_t> a_ip((size_t)n_rows + 1, 0);
    std::vector<int32_t> a_ix;
    for (int64_t r = 0; r < n_rows; ++r) {
        const int n = (r % 97 == 0) ? 0 : (int)(rng() % 300);
        std::vector<int32_t> row;
        for (int i = 0; i < n; ++i) row.push_back((int32_t)(rng() % (uint64_t)(F + (r % 1013 == 0 ? 7 : 0))));
        std::sort(row.begin(), row.end());
        a_ix.insert(a_ix.end(), row.begin(), row.end());
        a_ip[(size_t)r + 1] = (int64_t)a_ix.size();
    }
    std::vector<int64_t> want_ip((size_t)n_rows + 1, 0);
    std::vector<int32_t> want_ix;
    for (int64_t r = 0; r < n_rows; ++r) {
        for (int64_t q = a_ip[(size_t)r]; q < a_ip[(size_t)r + 1]; ++q) {
            const int64_t f = a_ix[(size_t)q];
            if (f < F && indptr[(size_t)f + 1] > indptr[(size_t)f]) want_ix.push_back((int32_t)f);
        }
        want_ip[(size_t)r + 1] = (int64_t)want_ix.size();
    }
    for (int threads : {1, 5}) {
        std::vector<int64_t> o_ip((size_t)n_rows + 1);
        std::vector<int32_t> o_ix(want_ix.size());  // exactly enough room: one more write would be caught
        rc = csrc::compact(bits, F, n_rows, a_ip.data(), a_ix.data(), o_ip.data(), o_ix.data(), (int64_t)o_ix.size(), threads);
        if (rc || o_ip != want_ip || o_ix != want_ix) return printf("FAIL compact threads=%d rc=%d\n", threads, rc), 1;
        // the chunk form of the pipelined upload (host_upload.inc; AVX-512 rows where the CPU has them): any row range
        for (int64_t c0 = 0; c0 < n_rows; c0 += 97) {
            const int64_t c1 = std::min<int64_t>(n_rows, c0 + 97), raw = a_ip[(size_t)c1] - a_ip[(size_t)c0];
            std::vector<int32_t> buf((size_t)raw + 16);
            std::vector<int64_t> ptr((size_t)(c1 - c0) + 1);
            const int64_t n = csrc::compact_chunk(bits.data(), (uint64_t)F, a_ip.data(), a_ix.data(), c0, c1, buf.data(), raw, ptr.data());
            if (n != want_ip[(size_t)c1] - want_ip[(size_t)c0]) return printf("FAIL compact_chunk count at row %lld\n", (long long)c0), 1;
            for (int64_t r = c0; r <= c1; ++r)
                if (ptr[(size_t)(r - c0)] != want_ip[(size_t)r] - want_ip[(size_t)c0]) return printf("FAIL compact_chunk pointers\n"), 1;
            if (n > 0 && memcmp(buf.data(), want_ix.data() + want_ip[(size_t)c0], (size_t)n * 4) != 0) return printf("FAIL compact_chunk ids\n"), 1;
            if (raw > 0 && csrc::compact_chunk(bits.data(), (uint64_t)F, a_ip.data(), a_ix.data(), c0, c1, buf.data(), raw - 1, ptr.data()) != -1 &&
                n == raw)
                return printf("FAIL compact_chunk capacity\n"), 1;
        }
        if (!want_ix.empty() &&
            csrc::compact(bits, F, n_rows, a_ip.data(), a_ix.data(), o_ip.data(), o_ix.data(), (int64_t)o_ix.size() - 1, threads) == FDR_OK)
            return printf("FAIL capacity\n"), 1;
    }
    printf("rc=0 rows=%u nnz=%zu kept=%zu of %zu\n", T.rows, cols.size(), want_ix.size(), a_ix.size());
    return 0;
}

static int check_plan(int cus, int64_t nq, int64_t nt, int d, int k, int shape) {
    const KnnPlan p = knn_plan(cus, nq, nt, d, k, shape);
    const KnnShape &sh = kShapes[p.shape];
    const int T = (int)((nt + 31) / 32);
    bool ok = p.nseg >= 1 && p.nseg <= FDR_MAX_SEG && p.segs.b[0] == 0 && p.qw == 32 * sh.nq * sh.nw &&
              (int64_t)p.nqb * p.qw >= nq && p.nq_pad == p.nqb * p.qw;
    for (int i = 0; i < p.nseg && ok; ++i) {
        const int len = p.segs.b[i + 1] - p.segs.b[i];
        ok = len > 0 && len % 32 == 0 && (sh.tps == 0 || len <= (1 << FDR_PREFILTER_MAX_IB));
    }
    ok = ok && p.segs.b[p.nseg] == T * 32;
    for (int i = p.nseg; i <= FDR_MAX_SEG && ok; ++i) ok = p.segs.b[i] == T * 32;
    ok = ok && p.cohort >= 0 && (p.cohort == 0 || (p.cohort <= cus * 4 && sh.tps > 0)) &&
         p.total_bytes == p.bits_bytes + p.shared_bytes + p.partial_bytes &&
         p.partial_bytes == (size_t)p.nseg * p.nq_pad * (size_t)k * 8;
    if (!ok) printf("FAIL plan cus=%d nq=%lld nt=%lld d=%d k=%d shape=%d nseg=%d\n", cus, (long long)nq, (long long)nt, d, k, shape, p.nseg);
    return ok ? 0 : 1;
}

static int cmd_plan() {
    int bad = 0, n = 0;
    const int64_t sizes[] = {20, 64, 8191, 8192, 8193, 100000, (1 << 19) - 1, 1 << 19, (1 << 19) + 1, 1000000,
                             1250000, 2500000, 10000000, 20000000, ((int64_t)FDR_MAX_SEG << FDR_PREFILTER_MAX_IB)};
    for (int cus : {256, 304, 64})
        for (int64_t nt : sizes)
            for (int d : {16, 128, 256, 500})
                for (int k : {1, 20, 50, 64}) {
                    if (nt < k) continue;
                    const int dp = padded_dim(d);
                    for (int64_t nq : {nt, (nt + 7) / 8, (int64_t)1}) {
                        bad += check_plan(cus, nq, nt, d, k, -1);  // exact shapes
                        ++n;
                        const int kp = (k + prefilter_extra(k) + 1) & ~1;
                        if (kp <= FDR_FAST_MAX_K && nt >= kp) {
                            bad += check_plan(cus, nq, nt, d, kp, prefilter_shape(dp, kp, nq, cus));
                            bad += check_plan(cus, nq, nt, d, 1, range_shape(dp));
                            n += 2;
                        }
                    }
                }
    // configs 4 / 5 of BASELINE.json: one rank's plan must fit FDR_MAX_SEG segments
    const KnnPlan c4 = knn_plan(256, 1250000, 10000000, 128, 28, prefilter_shape(128, 28, 1250000, 256));
    const KnnPlan c5 = knn_plan(256, 2500000, 20000000, 256, 58, prefilter_shape(256, 58, 2500000, 256));
    printf("rc=%d plans=%d config4_nseg=%d config5_nseg=%d\n", bad ? 1 : 0, n, c4.nseg, c5.nseg);
    return bad ? 1 : 0;
}

int main(int argc, char **argv) {
    const std::string cmd = argc > 1 ? argv[1] : "";
    if (cmd == "loader" && argc == 5) return cmd_loader(argv[2], atoll(argv[3]), atoi(argv[4]), false);
    if (cmd == "loader-stale" && argc == 4) return cmd_loader(argv[2], atoll(argv[3]), 2, true);
    if (cmd == "loader-range" && argc == 8)
        return cmd_loader_range(argv[2], atoll(argv[3]), atoi(argv[4]), atoll(argv[5]), atoll(argv[6]), atoi(argv[7]));
    if (cmd == "reads" && argc == 6) return cmd_reads(argv[2], atoll(argv[3]), atoi(argv[4]), argv[5]);
    if (cmd == "tables" && argc == 5) return cmd_tables((unsigned)atoi(argv[2]), atoll(argv[3]), atoi(argv[4]));
    if (cmd == "plan") return cmd_plan();
    if (cmd == "floats" && argc == 3) {  // host_san floats N: float32 bit patterns (one hex word per line on stdin) -> text
        char line[64], out[64];
        long long n = atoll(argv[2]);
        while (n-- > 0 && fgets(line, sizeof(line), stdin)) {
            const uint32_t bits = (uint32_t)strtoul(line, nullptr, 16);
            float x;
            memcpy(&x, &bits, 4);
            const int len = ovw::format_float32(x, out);
            fwrite(out, 1, (size_t)len, stdout);
            fputc('\n', stdout);
        }
        return 0;
    }
    if (cmd == "overlaps" && argc == 4) {  // host_san overlaps OUT THREADS: a random graph incl. -1 fillers, self hits, inf
        std::mt19937_64 rng(7);
        const int64_t n = 3000;
        const int k = 11;
        std::vector<int64_t> off((size_t)n + 1, 0);
        std::string names;
        std::vector<uint8_t> strands((size_t)n);
        for (int64_t i = 0; i < n; ++i) {
            names += "read_" + std::to_string(i / 2) + std::string((size_t)(rng() % 5), 'x');
            if (i % 211 == 0) names += "\"q\"";  // (a name csv.QUOTE_MINIMAL quotes and doubles)
            if (i % 307 == 0) names += "\tz";
            off[(size_t)i + 1] = (int64_t)names.size();
            strands[(size_t)i] = (uint8_t)(i & 1);
        }
        std::vector<int32_t> idx((size_t)(n * k));
        std::vector<float> dist((size_t)(n * k));
        for (int64_t i = 0; i < n * k; ++i) {
            idx[(size_t)i] = (rng() % 50 == 0) ? -1 : (int32_t)(rng() % (uint64_t)n);
            const uint32_t bits = (uint32_t)(rng() % 0x3f800001u);  // [0, 1]
            memcpy(&dist[(size_t)i], &bits, 4);
            if (rng() % 97 == 0) dist[(size_t)i] = std::numeric_limits<float>::infinity();
            if (rng() % 389 == 0) dist[(size_t)i] = std::numeric_limits<float>::quiet_NaN();  // (an empty field)
        }
        for (int64_t q = 0; q < n; q += 3) idx[(size_t)(q * k)] = (int32_t)q;
        int64_t lines = 0, lines2 = 0;
        // whole graph in one call, then the same graph as two row blocks appended to a second file
        int rc = fdr_overlaps_write(argv[2], 0, 1, n, 0, n, k, idx.data(), dist.data(), off.data(), names.data(),
                                    strands.data(), atoi(argv[3]), &lines);
        const std::string p2 = std::string(argv[2]) + ".parts";
        int64_t l1 = 0;
        if (!rc) rc = fdr_overlaps_write(p2.c_str(), 0, 1, n, 0, 1000, k, idx.data(), dist.data(), off.data(), names.data(),
                                         strands.data(), atoi(argv[3]), &l1);
        if (!rc) rc = fdr_overlaps_write(p2.c_str(), 1, 0, n, 1000, n - 1000, k, idx.data() + 1000 * k, dist.data() + 1000 * k,
                                         off.data(), names.data(), strands.data(), atoi(argv[3]), &lines2);
        idx[5] = (int32_t)n;  // out of range: refused
        const int rc_bad = fdr_overlaps_write((std::string(argv[2]) + ".bad").c_str(), 0, 1, n, 0, n, k, idx.data(), dist.data(),
                                              off.data(), names.data(), strands.data(), 1, nullptr);
        printf("rc=%d lines=%lld parts=%lld bad_rc=%d\n", rc, (long long)lines, (long long)(l1 + lines2), rc_bad);
        return rc ? 1 : 0;
    }
    if (cmd == "plan-print" && argc == 7) {  // host_san plan-print NQ NT D K SHAPE  (shape -1: the exact mode's choice)
        const KnnPlan p = knn_plan(256, atoll(argv[2]), atoll(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]));
        printf("shape=%d nqb=%d nseg=%d cohort=%d queues=%d tiles:", p.shape, p.nqb, p.nseg, p.cohort, p.queues);
        for (int i = 0; i < p.nseg; ++i) printf(" %d", (p.segs.b[i + 1] - p.segs.b[i]) / 32);
        printf(" bytes=%zu\n", p.total_bytes);
        return 0;
    }
    fprintf(stderr, "usage: host_san loader|loader-stale|tables|plan ...\n");
    return 2;
}
