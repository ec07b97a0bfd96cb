// snapgpu_sam.cpp -- FASTQ batcher + SAM writer over the C ABI: the host side of
//     snapgpu-sam single <index-dir> <reads.fq> -o <out.sam> [-d maxDist] [-G-] [-=] [-M] [-Cxx] [-ea] [-D n] [-om n [-omax n] [-mpc n]] [-mrl minReadLength]
//                        [-b readsPerBatch] [-gpus n] [-q contextsPerGpu] [-t formatterThreads]
//     snapgpu-sam paired <index-dir> <reads1.fq> <reads2.fq> -o <out.sam> [same options]
// Streams FASTQ records in batches across include/snapgpu.h -- snapgpu_align_single / snapgpu_align_paired (BaseAligner::AlignRead,
// ChimericPairedEndAligner::align) and snapgpu_sam_fields_single / _paired (what SimpleReadWriter::writeReads / writePairs compute before
// they print) -- and prints the records the way the reference does.
//
// Shape (the reference's: SNAPLib/ParallelTask.h:128-138 runs one aligner per thread over a shared read supplier, SingleAligner.cpp:197-330):
//     reader thread  ->  [batches]  ->  GPU feeder threads  ->  [aligned batches]  ->  formatter threads  ->  writer (batch order)
//   * one context per (GPU, feeder thread), `-q` feeders per GPU (default 2): while one feeder's batch is in the kernel the other's is being
//     copied in / out, so H2D, kernel and D2H of consecutive batches overlap on separate streams without a global lock; the feeders of one GPU
//     share its index blobs (snapgpu_create_replica(..., share_index = 1));
//   * `-gpus n` (default: every visible device): the index is read from disk ONCE, into GPU 0, and replicated into the other GPUs' HBM by
//     snapgpu_broadcast_index (RCCL broadcast over xGMI); batches go to whichever feeder is free, reads never cross GPUs;
//   * results are re-ordered by batch number before they are written, so the file is what a single-threaded run writes.
// What is restated here is host-side text handling only:
//   FASTQReader::getReadFromBuffer   SNAPLib/FASTQ.cpp:148-260   (4-line records, '\r' tolerated; plain or gzip input through zlib)
//   Read::clip (ClipBack)            SNAPLib/Read.h:567-620      (the CLI default -C-+: drop the trailing run of '#' qualities)
//   the "useless read" filter        SNAPLib/SingleAligner.cpp:211-232 (dataLength < -mrl or more Ns than -d: written unaligned)
//   SAMFormat::writeHeader           SNAPLib/SAM.cpp:1204-1305   (@HD, default @RG, @PG, one @SQ per contig in ORIGINAL FASTA order, :1291)
//   SAMFormat::writeRead's snprintf  SNAPLib/SAM.cpp:2078-2098   (field order, PG:Z:SNAP, NM:i, default read-group aux)
//   paired: the both-mates-useless rule (PairedAligner.cpp:680-707), the /1 /2 suffix rule (ReadWriter.cpp:392-404), the mate's quality
//   sum QS:i (SAM.cpp:1826-1837) and writePairs' snprintf (:1855-1877); pairing arithmetic is snapgpu_sam_fields_paired's
// No alignment arithmetic happens on the host: without a GPU snapgpu_create_from_directory fails and so does this program.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <chrono>
#include <vector>
#include <dlfcn.h>
#include <zlib.h>                        // gzread reads plain and gzip-compressed FASTQ alike (the reference takes .gz input too)

#include "../../../include/snapgpu.h"

// The output is written to "<out>.partial" and renamed when it is complete, so that a run that stops half way (an input error, one of the
// unsupported paired -om corners below) never leaves a truncated SAM -- or a BAM without its end-of-file block -- under the name asked for.
static std::string g_partial_path;
// where the host threads' time goes (SNAPGPU_SAM_VERBOSE=1 prints it): nanoseconds summed over the threads of a kind
static std::atomic<unsigned long long> g_ns_prep(0), g_ns_align(0), g_ns_mid(0), g_ns_samcall(0), g_ns_format(0), g_ns_write(0), g_ns_parse(0);
static std::atomic<unsigned long long> g_ns_wait_in(0), g_ns_wait_out(0), g_ns_wait_writer(0);      // feeders waiting for parsed batches / for room behind them; the writer waiting for the next batch in order
struct StageTimer {
    std::atomic<unsigned long long> &acc; std::chrono::steady_clock::time_point t0;
    explicit StageTimer(std::atomic<unsigned long long> &a) : acc(a), t0(std::chrono::steady_clock::now()) {}
    ~StageTimer() { acc += (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
};
static void die(const char *msg, const char *arg = "") {
    fprintf(stderr, "snapgpu-sam: %s%s\n", msg, arg);
    if (!g_partial_path.empty()) remove(g_partial_path.c_str());
    exit(1);
}

struct Contig { std::string name; uint64_t begin; bool is_alt; int orig; };

// the contig table of the index directory: names for RNAME / @SQ (Genome.cpp:203-229, 353-403; GenomeIndex.cpp:1879)
static void load_contigs(const std::string &dir, std::vector<Contig> &contigs, uint64_t &n_bases, uint32_t &padding)
{
    FILE *f = fopen((dir + "/GenomeIndex").c_str(), "r");
    if (!f) die("cannot open GenomeIndex in ", dir.c_str());
    unsigned major = 0, minor = 0, n_tables = 0, seed_len = 0, pad = 0; unsigned long long ovf = 0;
    if (fscanf(f, "%u %u %u %llu %u %u", &major, &minor, &n_tables, &ovf, &seed_len, &pad) != 6) die("malformed GenomeIndex header");
    fclose(f);
    padding = pad;
    f = fopen((dir + "/Genome").c_str(), "rb");
    if (!f) die("cannot open Genome in ", dir.c_str());
    char line[8192];
    long long nb = 0; int nc = 0;
    if (!fgets(line, sizeof(line), f) || sscanf(line, "%lld %d", &nb, &nc) != 2) die("malformed Genome header");
    n_bases = (uint64_t)nb;
    for (int i = 0; i < nc; i++) {
        long long begin = 0, pbegin = 0; int cflags = 0, orig = 0, pflags = 0, name_len = 0, cigar_len = 0, consumed = 0;
        if (!fgets(line, sizeof(line), f) ||
            sscanf(line, "%lld %x %d %lld %x %d %d %n", &begin, &cflags, &orig, &pbegin, &pflags, &name_len, &cigar_len, &consumed) < 7)
            die("malformed contig line in Genome");
        Contig c; c.begin = (uint64_t)begin; c.is_alt = (cflags & 1) != 0; c.orig = orig; c.name.assign(line + consumed, (size_t)name_len);
        contigs.push_back(c);
    }
    fclose(f);
}

// ---------------------------------------------------------------------------------------- FASTQ input
struct LineReader {                      // lines out of a gz / plain file through one large buffer (gzgets is several times slower)
    gzFile f = NULL;
    std::vector<char> buf; size_t pos = 0, end = 0; bool eof = false;
    void open(const char *path) { f = gzopen(path, "rb"); if (!f) die("cannot open ", path); gzbuffer(f, 1 << 20); buf.resize(8u << 20); }
    void close() { if (f) gzclose(f); f = NULL; }
    bool fill() {
        if (eof) return false;
        if (pos > 0) { memmove(buf.data(), buf.data() + pos, end - pos); end -= pos; pos = 0; }
        if (end == buf.size()) buf.resize(buf.size() * 2);
        const int n = gzread(f, buf.data() + end, (unsigned)(buf.size() - end));
        if (n < 0) die("read error in FASTQ input");
        if (n == 0) { eof = true; return false; }
        end += (size_t)n;
        return true;
    }
    // next line without its terminator ('\r' tolerated); false at end of file
    bool line(const char *&s, size_t &n) {
        for (;;) {
            const char *nl = (const char *)memchr(buf.data() + pos, '\n', end - pos);
            if (nl) {
                s = buf.data() + pos; n = (size_t)(nl - s); pos += n + 1;
                if (n > 0 && s[n - 1] == '\r') n--;
                return true;
            }
            if (!fill()) {
                if (pos == end) return false;
                s = buf.data() + pos; n = end - pos; pos = end;
                if (n > 0 && s[n - 1] == '\r') n--;
                return true;
            }
        }
    }
};

struct Batch {
    uint64_t seq = 0;
    std::vector<uint32_t> name_off;          // n + 1 offsets into names
    std::string names;
    std::vector<char> bases, quals;          // unclipped, concatenated
    std::vector<uint64_t> offsets;           // n + 1
    size_t n() const { return offsets.size() - 1; }
    void clear() { name_off.assign(1, 0); names.clear(); bases.clear(); quals.clear(); offsets.assign(1, 0); }
};

// one FASTQ record appended to the batch; false at end of file (blank lines between records are skipped)
static bool next_read(LineReader &in, Batch &b, uint32_t max_read_len)
{
    const char *s; size_t n;
    do { if (!in.line(s, n)) return false; } while (n == 0);
    if (s[0] != '@') die("FASTQ record does not start with '@': ", std::string(s, n).c_str());
    const size_t id_at = b.names.size();
    b.names.append(s + 1, n - 1); b.name_off.push_back((uint32_t)b.names.size());
    auto id = [&]() { return std::string(b.names.data() + id_at, b.names.size() - id_at); };      // (only built for an error message)
    if (!in.line(s, n)) die("truncated FASTQ record: ", id().c_str());
    const size_t len = n;
    if (len > max_read_len) die("read longer than max_read_len (400; set SNAPGPU_MAX_READ_LEN, at most 1000): ", id().c_str());
    b.bases.insert(b.bases.end(), s, s + n);
    if (!in.line(s, n)) die("truncated FASTQ record: ", id().c_str());
    if (n == 0 || s[0] != '+') die("FASTQ record without '+' line: ", id().c_str());
    if (!in.line(s, n)) die("truncated FASTQ record: ", id().c_str());
    if (n != len) die("FASTQ sequence and quality lengths differ: ", id().c_str());
    b.quals.insert(b.quals.end(), s, s + n);
    b.offsets.push_back(b.bases.size());
    return true;
}

// ---------------------------------------------------------------------------------------- plain FASTQ, read by many threads
// A plain (not gzip) FASTQ file is mapped and indexed by line count -- one pass, the chunks counted side by side -- so that record r is line
// 4 r and any thread can parse any batch of records on its own (FASTQ.h:67 / ReadSupplierQueue.h:76: the reference's reader hands buffers
// to a queue of supplier threads for the same reason).  A file whose lines do not come in fours (blank lines between records, which the
// sequential reader skips) is read by the sequential reader instead; -seqread forces that reader.
struct MappedFastq {
    const char *p = NULL; size_t n = 0;
    static const size_t CHUNK = 1u << 20;
    std::vector<uint64_t> lines_before;       // lines that start before chunk c (c = 0 .. n_chunks)
    uint64_t n_lines = 0;
    bool open(const char *path) {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 2) { ::close(fd); return false; }
        unsigned char magic[2] = {0, 0};
        if (pread(fd, magic, 2, 0) != 2 || (magic[0] == 0x1f && magic[1] == 0x8b)) { ::close(fd); return false; }      // gzip: the sequential reader
        void *m = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        ::close(fd);
        if (m == MAP_FAILED) return false;
        madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
        p = (const char *)m; n = (size_t)st.st_size;
        return true;
    }
    void close() { if (p) munmap((void *)p, n); p = NULL; }
    // count the newlines of every chunk with `threads` threads; false when the lines do not come in fours
    bool index(int threads) {
        const size_t n_chunks = (n + CHUNK - 1) / CHUNK;
        std::vector<uint64_t> cnt(n_chunks, 0);
        std::atomic<size_t> next(0);
        std::vector<std::thread> ts;
        for (int t = 0; t < threads; t++)
            ts.emplace_back([&] {
                for (;;) {
                    const size_t c = next.fetch_add(1);
                    if (c >= n_chunks) break;
                    const char *q = p + c * CHUNK, *e = p + (c + 1 == n_chunks ? n : (c + 1) * CHUNK);
                    uint64_t k = 0;
                    while (q < e) { const char *nl = (const char *)memchr(q, '\n', (size_t)(e - q)); if (!nl) break; k++; q = nl + 1; }
                    cnt[c] = k;
                }
            });
        for (auto &t : ts) t.join();
        lines_before.assign(n_chunks + 1, 0);
        for (size_t c = 0; c < n_chunks; c++) lines_before[c + 1] = lines_before[c] + cnt[c];
        n_lines = lines_before[n_chunks] + (p[n - 1] != '\n' ? 1 : 0);          // (a last line without its newline)
        return n_lines % 4 == 0;
    }
    // byte offset of the start of line L (0-based; L < n_lines)
    size_t line_start(uint64_t L) const {
        if (L == 0) return 0;
        // the chunk in which newline number L (1-based) lies: lines_before[c] < L <= lines_before[c + 1]
        size_t lo = 0, hi = lines_before.size() - 1;
        while (hi - lo > 1) { const size_t mid = (lo + hi) / 2; if (lines_before[mid] < L) lo = mid; else hi = mid; }
        const char *q = p + lo * CHUNK, *e = p + n;
        uint64_t k = lines_before[lo];
        while (q < e) { const char *nl = (const char *)memchr(q, '\n', (size_t)(e - q)); if (!nl) break; q = nl + 1; if (++k == L) return (size_t)(q - p); }
        return n;
    }
};

// `count` records starting at byte `at` of a mapped file, appended to the batch (every `step`-th slot when two files are interleaved is the
// caller's business: paired batches call this per record).  Returns the byte after the last record.
static inline size_t parse_mapped_record(const MappedFastq &f, size_t at, Batch &b, uint32_t max_read_len)
{
    const char *p = f.p, *e = f.p + f.n;
    auto line = [&](const char *&s, size_t &n) -> bool {
        if (at >= f.n) return false;
        const char *q = p + at;
        const char *nl = (const char *)memchr(q, '\n', (size_t)(e - q));
        s = q; n = nl ? (size_t)(nl - q) : (size_t)(e - q); at += n + (nl ? 1 : 0);
        if (n > 0 && s[n - 1] == '\r') n--;
        return true;
    };
    const char *s; size_t n;
    if (!line(s, n)) die("truncated FASTQ file");
    if (n == 0 || s[0] != '@') die("FASTQ record does not start with '@' (blank lines between records? -seqread reads such files): ", std::string(s, n < 80 ? n : 80).c_str());
    const size_t id_at = b.names.size();
    b.names.append(s + 1, n - 1); b.name_off.push_back((uint32_t)b.names.size());
    auto id = [&]() { return std::string(b.names.data() + id_at, b.names.size() - id_at); };
    if (!line(s, n)) die("truncated FASTQ record: ", id().c_str());
    const size_t len = n;
    if (len > max_read_len) die("read longer than max_read_len (400; set SNAPGPU_MAX_READ_LEN, at most 1000): ", id().c_str());
    b.bases.insert(b.bases.end(), s, s + n);
    if (!line(s, n)) die("truncated FASTQ record: ", id().c_str());
    if (n == 0 || s[0] != '+') die("FASTQ record without '+' line: ", id().c_str());
    if (!line(s, n)) die("truncated FASTQ record: ", id().c_str());
    if (n != len) die("FASTQ sequence and quality lengths differ: ", id().c_str());
    b.quals.insert(b.quals.end(), s, s + n);
    b.offsets.push_back(b.bases.size());
    return at;
}

// ---------------------------------------------------------------------------------------- options, shared state
struct Options {
    bool paired = false;
    snapgpu_params p;
    snapgpu_paired_params pp;
    bool use_m = true;                                                     // AlignerOptions.cpp:58
    unsigned min_read_len = 50;                                            // -mrl
    size_t batch_reads = 0;                                                // -b; 0 = auto (main)
    size_t group = 0;                                                      // -g: single-end batches per GPU call (0 = auto: up to ~1 M reads)
    bool clip_front = false, clip_back = true;                             // -C-+ (ClipBack) is the default
    int om = -1, mpc = -1; long long omax = 0x7fffffff;
    bool ae = false;                                                       // -ae: AlignmentAdjuster before the -om filter (single end only)
    bool stop_on_first_hit = false, explore_popular_seeds = false;         // -f, -x (single end; the paired-end aligners ignore them, as the reference's do)
    int n_gpus = 0, ctx_per_gpu = 0, n_format = 0, n_parse = 0;
    bool seqread = false;                                                  // -seqread: the sequential FASTQ reader even for a plain file
    int passes = 1;                                                        // -passes N (measurement): stream the input N times over the resident index
    uint32_t ops_stride = 64;
    bool bam = false;                                                      // -o x.bam: BAM records in BGZF blocks (SNAPLib/Bam.cpp)
};

struct Work {                                // a batch on its way through the pipeline
    Batch b;
    // single-end: one entry per RECORD (a read with secondary results appears once per record); paired: one per read
    std::vector<uint32_t> rec_read; std::vector<char> rec_secondary;
    std::vector<int32_t> flag, contig, mapq, n_ops, nm, rnext, first_written;
    std::vector<int64_t> pos, pnext, tlen;
    std::vector<uint32_t> ops; uint32_t ops_stride = 64;
    // paired end: what to write, in order.  A "pair unit" is one PairedAlignmentResult of a pair (the primary, a secondary one, the first
    // ALT one): two records, fields at [2u], [2u + 1] of the arrays above, first_written[u].  A "single unit" is one single-end secondary
    // result of a mate (ReadWriter.cpp:508-590): one record, fields in the s_* arrays.
    struct Emit { uint32_t unit; uint8_t single; };
    std::vector<Emit> emit;
    std::vector<uint32_t> pu_pair; std::vector<char> pu_secondary;       // per pair unit: the pair it belongs to, flag 0x100
    std::vector<uint32_t> su_read;                                       // per single unit: the read (2 * pair + mate)
    std::vector<int32_t> s_flag, s_contig, s_mapq, s_n_ops, s_nm; std::vector<int64_t> s_pos; std::vector<uint32_t> s_ops; uint32_t s_ops_stride = 64;
    // single end: what prepare_single() leaves for the feeder
    bool prepared = false;
    std::vector<int32_t> front_clip, data_len; std::vector<uint32_t> to_align; std::vector<char> ab, aq; std::vector<uint64_t> ao;
    std::vector<uint8_t> skip;               // single end: read i is not given to the aligner (the fused call's argument)
    uint32_t max_len = 0;                    // the longest read of the batch (unclipped): picks the context's read-length class
    std::string text;                        // the formatted records (BAM: BGZF blocks)
    bool bam = false;
    unsigned long long mapped = 0;
    // as newly constructed, but every vector keeps its capacity (WorkPool)
    void reset() {
        b.clear(); b.seq = 0;
        rec_read.clear(); rec_secondary.clear(); flag.clear(); contig.clear(); mapq.clear(); n_ops.clear(); nm.clear(); rnext.clear(); first_written.clear();
        pos.clear(); pnext.clear(); tlen.clear(); ops.clear(); ops_stride = 64;
        emit.clear(); pu_pair.clear(); pu_secondary.clear(); su_read.clear();
        s_flag.clear(); s_contig.clear(); s_mapq.clear(); s_n_ops.clear(); s_nm.clear(); s_pos.clear(); s_ops.clear(); s_ops_stride = 64;
        prepared = false; front_clip.clear(); data_len.clear(); to_align.clear(); ab.clear(); aq.clear(); ao.clear(); skip.clear();
        max_len = 0; text.clear(); bam = false; mapped = 0;
    }
};

// Batches are RECYCLED: a batch of 131 072 reads is ~100 MB of vectors (read text, record fields, cigar operations, the formatted records), and allocating
// them anew for every batch meant ~15 GB of fresh pages per 20 M reads -- page faults in the parser / feeder / formatter threads and an munmap per vector
// when the batch was freed, all serialised on the process's address-space lock (freeing on the writer's thread cost half a pass; freeing on a thread of its
// own slowed the feeders' calls by a half instead: profiles/r06g, r06h).  A recycled batch is as newly constructed except for its vectors' capacity.
struct WorkPool {
    std::mutex m; std::vector<Work *> free_list; size_t cap;
    explicit WorkPool(size_t c) : cap(c) {}
    Work *get() { { std::lock_guard<std::mutex> l(m); if (!free_list.empty()) { Work *w = free_list.back(); free_list.pop_back(); return w; } } return new Work(); }
    void put(Work *w) { w->reset(); { std::lock_guard<std::mutex> l(m); if (free_list.size() < cap) { free_list.push_back(w); return; } } delete w; }
    ~WorkPool() { for (Work *w : free_list) delete w; }
};

template <class T> struct Queue {            // bounded multi-producer multi-consumer queue; close() lets consumers drain and stop
    std::mutex m; std::condition_variable not_empty, not_full; std::deque<T> q; size_t cap; bool closed = false;
    explicit Queue(size_t c) : cap(c) {}
    void push(T v) { std::unique_lock<std::mutex> l(m); not_full.wait(l, [&] { return q.size() < cap; }); q.push_back(std::move(v)); not_empty.notify_one(); }
    bool pop(T &v) {
        std::unique_lock<std::mutex> l(m); not_empty.wait(l, [&] { return !q.empty() || closed; });
        if (q.empty()) return false;
        v = std::move(q.front()); q.pop_front(); not_full.notify_one(); return true;
    }
    // up to `want` items: waits for the first; takes what else is there, and waits a little (`grace`) for the rest while producers are alive
    size_t pop_upto(std::vector<T> &out, size_t want, std::chrono::milliseconds grace) {
        std::unique_lock<std::mutex> l(m); not_empty.wait(l, [&] { return !q.empty() || closed; });
        const auto until = std::chrono::steady_clock::now() + grace;
        for (;;) {
            while (!q.empty() && out.size() < want) { out.push_back(std::move(q.front())); q.pop_front(); not_full.notify_one(); }
            if (out.size() >= want || closed || out.empty()) break;
            if (not_empty.wait_until(l, until, [&] { return !q.empty() || closed; }) == false) break;
        }
        return out.size();
    }
    void close() { std::lock_guard<std::mutex> l(m); closed = true; not_empty.notify_all(); }
};

static char complement(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N'; }      // COMPLEMENT[], Tables.cpp

static inline void put_uint(std::string &o, unsigned long long v) { char t[24]; int n = 0; do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v); while (n) o.push_back(t[--n]); }
static inline void put_int(std::string &o, long long v) { if (v < 0) { o.push_back('-'); put_uint(o, (unsigned long long)(-(v + 1)) + 1ull); } else put_uint(o, (unsigned long long)v); }

static void fail_rc(snapgpu_ctx *ctx, const char *what, int rc) { fprintf(stderr, "snapgpu-sam: %s failed (%d): %s\n", what, rc, snapgpu_last_error(ctx)); exit(1); }

// Read::clip and the useless-read test for read i of the batch (Read.h:586-608: back first, then front; SingleAligner.cpp:211-232)
static inline bool clip_read(const Options &o, const Batch &b, size_t i, int32_t &front_clip, int32_t &data_len)
{
    const char *q = b.quals.data() + b.offsets[i], *s = b.bases.data() + b.offsets[i];
    size_t m = (size_t)(b.offsets[i + 1] - b.offsets[i]), fc = 0;
    if (o.clip_back) while (m > 0 && q[m - 1] == '#') m--;
    if (o.clip_front) while (fc < m && q[fc] == '#') fc++;
    m -= fc;
    front_clip = (int32_t)fc; data_len = (int32_t)m;
    unsigned n_count = 0;
    for (size_t j = 0; j < m; j++) n_count += s[fc + j] == 'N';
    return m >= o.min_read_len && n_count <= o.p.max_k;
}

// ---------------------------------------------------------------------------------------- the record loop on the host (rare path)
// SAMFormat::writePairs / writeRead's attempt loop (SAM.cpp:1630-1721, ReadWriter.cpp:228-311 / :508-590) with the Read's clipping state as an
// INPUT: what sam_fields.h does on the device for a record that starts from a fresh Read, done here for the records that do not -- a later
// result of a read whose earlier record left additional back clipping behind (Read.h:547-553).  One snapgpu_compute_cigar_* call per attempt;
// the arithmetic is sam_fields.h's (createSAMLine, computeCigarString, getRefSpanFromCigar, fillMateInfo), line for line.
static const std::vector<Contig> *g_contigs = NULL; static uint64_t g_n_bases = 0; static uint32_t g_padding = 0;
static int h_contig_at(long long loc) { int lo = 0, hi = (int)g_contigs->size() - 1, c = -1; while (lo <= hi) { const int mid = (lo + hi) >> 1; if ((long long)(*g_contigs)[(size_t)mid].begin <= loc) { c = mid; lo = mid + 1; } else hi = mid - 1; } return c; }
static long long h_contig_end(int c) { return c == (int)g_contigs->size() - 1 ? (long long)g_n_bases : (long long)(*g_contigs)[(size_t)c + 1].begin; }
struct HostRec {
    int flag = 0, contig = -1, mapq = 0, n_ops = -1, nm = -1; long long pos = 0;
    long long final_loc = -1; int final_dir = 0, bases_clipped_before = 0, ref_span = 0, data_len = 0;
    std::vector<uint32_t> ops;
    int back_after = 0;                      // the Read's additional back clipping once the record is written
};
struct HostRes { int status, direction, score, mapq, adj, used_ag, bcb, bca, supplementary; long long location; };

static HostRec host_record(const Options &o, snapgpu_ctx *ctx, const char *bases, const char *quals, int U, int F0, int D0, int back0, const HostRes &res, bool paired)
{
    HostRec out; out.back_after = back0;
    int addF = res.adj, addB = back0;
    int status = res.status, direction = res.direction;
    long long location = status == SNAPGPU_NotFound ? -1 : res.location, final_loc = location;
    const bool ag_branch = o.p.use_affine_gap && (res.used_ag != 0 || res.score > 0);
    int cum = 0, n_adj = 0;
    std::vector<char> od((size_t)U), oq((size_t)U);
    int oriented_dir = -1;
    uint32_t stride = 256;
    for (int attempt = 0; attempt < 2 * (int)o.p.max_read_len + 8; attempt++) {
        const int front = F0 + addF, dlen = D0 - addF - addB;
        long long loc = final_loc;
        if (status == SNAPGPU_NotFound) loc = -1;
        const int dir = loc < 0 ? 0 : direction;
        int clipped_len = dlen, bcb, bca;
        if (dir == 1) { bcb = U - clipped_len - front; bca = front; } else { bcb = front; bca = U - clipped_len - bcb; }
        if (ag_branch || paired) { bcb += res.bcb; bca += res.bca; clipped_len -= res.bcb + res.bca; }
        int flag = res.supplementary ? 0x800 : 0, contig = -1, mapq = 0;
        long long pos = 0, extra = 0;
        if (loc >= 0) {
            if (dir == 1) flag |= 0x10;
            contig = h_contig_at(loc);
            if (contig < 0 || loc + dlen > h_contig_end(contig)) {
                contig = contig + 1; if (contig >= (int)g_contigs->size()) contig = (int)g_contigs->size() - 1;
                extra = (long long)(*g_contigs)[(size_t)contig].begin - loc;
            }
            pos = loc + extra - (long long)(*g_contigs)[(size_t)contig].begin + 1;
            mapq = res.mapq < 0 ? 0 : (res.mapq > 70 ? 70 : res.mapq);
        } else flag |= 0x4;
        int afc = 0, nm = -1, n_ops = -1; bool star = true;
        long long clip_before = 0, clip_after = 0;
        std::vector<uint32_t> ops;
        if (ag_branch && !paired && extra != 0) afc = (int)extra;
        else if (loc >= 0) {
            if (oriented_dir != dir) {
                for (int i = 0; i < U; i++) { if (dir == 1) { od[(size_t)(U - 1 - i)] = complement(bases[i]); oq[(size_t)(U - 1 - i)] = quals[i]; } else { od[(size_t)i] = bases[i]; oq[(size_t)i] = quals[i]; } }
                oriented_dir = dir;
            }
            if (clipped_len < 0 || bcb < 0 || bcb + clipped_len > U) die("paired -om / -ea: clipping arithmetic out of range in the host record loop");
            const uint64_t off[2] = {0, (uint64_t)clipped_len};
            const int32_t len = clipped_len, xb = (int32_t)extra, sc = res.score; const int64_t l64 = loc;
            int32_t r_nops = -1, r_ed = -1, r_afc = 0, r_tail = 0, r_hist = 0; int64_t r_xa = 0;
            for (;;) {
                ops.assign(stride, 0);
                int rc = ag_branch
                    ? snapgpu_compute_cigar_ag(ctx, 1, od.data() + bcb, oq.data() + bcb, (uint64_t)clipped_len, off, &len, &l64, &xb, &sc, o.use_m ? 1 : 0, ops.data(), stride, &r_nops, &r_ed, &r_afc, &r_xa, &r_tail, &r_hist)
                    : snapgpu_compute_cigar_lv(ctx, 1, od.data() + bcb, (uint64_t)clipped_len, off, &len, &l64, &xb, o.use_m ? 1 : 0, ops.data(), stride, &r_nops, &r_ed, &r_afc, &r_xa);
                if (rc != SNAPGPU_OK) fail_rc(ctx, "snapgpu_compute_cigar", rc);
                if (r_nops >= 0 || r_ed != -2 || stride >= 4096) break;          // (-2: the ops did not fit the stride)
                stride *= 4;
            }
            n_ops = r_nops; afc = r_afc;
            if (afc == 0 || n_ops < 0) {
                afc = n_ops < 0 ? 0 : afc;
                nm = n_ops < 0 ? 0 : r_ed;
                if (n_ops >= 0 && r_ed >= 0) { star = false; if (ag_branch) bca += r_tail; clip_before = bcb + extra; clip_after = bca + r_xa; }
            }
        }
        if (afc == 0) {
            out.flag = flag; out.contig = loc >= 0 ? contig : -1; out.pos = pos; out.mapq = mapq; out.nm = nm;
            out.final_loc = loc; out.final_dir = dir; out.bases_clipped_before = bcb; out.data_len = dlen; out.back_after = addB;
            if (loc < 0 || star) { out.n_ops = -1; return out; }
            if (clip_before > 0) out.ops.push_back(((uint32_t)clip_before << 4) | 4u);
            out.ops.insert(out.ops.end(), ops.begin(), ops.begin() + n_ops);
            if (clip_after > 0) out.ops.push_back(((uint32_t)clip_after << 4) | 4u);
            out.n_ops = (int)out.ops.size();
            int span = 0;
            for (int i = 0; i < out.n_ops; i++) { const uint32_t code = out.ops[(size_t)i] & 15u; if (i == 0 ? (code != 4u && code != 5u) : (code != 1u)) span += (int)(out.ops[(size_t)i] >> 4); }
            out.ref_span = span;
            return out;
        }
        n_adj++;
        if (paired) {
            const int co = h_contig_at(final_loc), cn = h_contig_at(final_loc + afc);
            if (cn != co || cn < 0 || final_loc + afc > h_contig_end(co) - (long long)g_padding || n_adj > 2 * (int)o.p.max_read_len) { status = SNAPGPU_NotFound; location = -1; direction = 0; final_loc = -1; continue; }
            if (ag_branch) { if (afc < 0) { cum += afc; if (direction == 0) addF = -cum; else addB = -cum; } else final_loc += afc; }
            else { if (afc > 0) { cum += afc; addF = cum; } final_loc += afc; }
            continue;
        }
        const int c_orig = status == SNAPGPU_NotFound ? -1 : h_contig_at(location), c_new = status == SNAPGPU_NotFound ? -1 : h_contig_at(location + afc);
        const int c_lim = ag_branch ? c_new : c_orig;
        bool give_up = c_new < 0 || c_new != c_orig || n_adj > dlen;
        if (!give_up) give_up = final_loc + afc > h_contig_end(c_lim) - (long long)g_padding;
        if (give_up) { status = SNAPGPU_NotFound; location = -1; direction = 0; final_loc = -1; continue; }
        if (ag_branch) { if (afc < 0) { cum += afc; if (direction == 0) addF = -cum; else addB = -cum; } else final_loc = location + afc; }
        else { if (afc > 0) { cum += afc; addF = cum; } final_loc += afc; }
    }
    out.flag = 0x4; out.n_ops = -1; out.nm = -1; out.back_after = addB;
    return out;
}

// SAMFormat::fillMateInfo (SAM.cpp:1308-1421) from the two mates' own records; rnext: -1 "*", -2 "=", otherwise a contig index
static void host_mate_info(const HostRec &me, const HostRec &mate, bool first_in_pair, bool aligned_as_pair, int &flag, int &contig, long long &pos, int &rnext, long long &pnext, long long &tlen)
{
    flag = me.flag | 0x1 | (first_in_pair ? 0x40 : 0x80); contig = me.contig; pos = me.pos; rnext = -1; pnext = 0; tlen = 0;
    auto contig_for_read = [&](long long loc, int data_len, long long *extra) -> int {
        int c = h_contig_at(loc); *extra = 0;
        if (c < 0 || loc + data_len > h_contig_end(c)) { c = c + 1; if (c >= (int)g_contigs->size()) c = (int)g_contigs->size() - 1; *extra = (long long)(*g_contigs)[(size_t)c].begin - loc; }
        return c;
    };
    long long mate_loc = mate.final_loc, mate_extra = 0; bool rnext_eq = false;
    if (mate_loc >= 0) {
        const int mc = contig_for_read(mate_loc, mate.data_len, &mate_extra);
        mate_loc += mate_extra;
        rnext = mc; pnext = mate_loc - (long long)(*g_contigs)[(size_t)mc].begin + 1;
        if (mate.final_dir == 1) flag |= 0x20;
        if (me.final_loc < 0) { contig = mc; rnext_eq = true; pos = pnext; }
    } else { flag |= 0x8; rnext_eq = true; pnext = pos; }
    if (me.final_loc >= 0 && mate.final_loc >= 0) {
        if (aligned_as_pair) flag |= 0x2;
        long long extra = 0;
        const int c = contig_for_read(me.final_loc, me.data_len, &extra);
        const long long loc = me.final_loc + extra;
        const long long my_start = loc - me.bases_clipped_before - extra, my_end = loc + me.ref_span;
        const long long mate_start = mate_loc - mate.bases_clipped_before - mate_extra, mate_end = mate_loc + mate.ref_span;
        contig = c;
        if (my_start < mate_start) { if (me.final_dir == 0) tlen = mate.final_dir == 1 ? mate_end - my_start : mate_start - my_start; else tlen = mate.final_dir == 0 ? mate_start - my_end : mate_end - my_end; }
        else { if (me.final_dir == 1) tlen = mate.final_dir == 0 ? -(my_end - mate_start) : -(my_end - mate_end); else tlen = mate.final_dir == 0 ? -(my_start - mate_start) : -(my_start - mate_end); }
    }
    if (rnext_eq || (rnext >= 0 && rnext == contig)) rnext = -2;
}

// ---------------------------------------------------------------------------------------- GPU stage
// Contexts of one feeder thread, by READ-LENGTH CLASS.  The aligner's answers do not depend on the buffer size it was built for
// (BaseAligner's maxReadSize only sizes its arrays), but the kernel variant does: reads up to 160 bp run the 192-position affine-gap
// variant at six waves per SIMD, up to 256 bp the 256-position one, the command line's maximum (400 by default, as the reference's
// MAX_READ_LENGTH) the 384-position one at four -- which a run of 150 bp reads used to pay for (profiles/r05a: k_align_single<6> was 76 %
// of the GPU time of FASTQ -> SAM).  The class-0 context of every feeder exists from the start; a longer batch makes its feeder create
// the next class over the same resident index (snapgpu_create_replica_with_params), once.
static const uint32_t RL_CLASS[3] = {160, 256, 0};                       // (0: the command line's maximum)
struct FeederCtx {
    snapgpu_ctx *c[3] = {NULL, NULL, NULL};
    snapgpu_ctx *owner = NULL; int device = 0;                             // whose index blobs they share
};
static uint32_t class_len(const Options &o, int k) { return RL_CLASS[k] && RL_CLASS[k] < o.p.max_read_len ? RL_CLASS[k] : o.p.max_read_len; }
static void configure_ctx(const Options &o, snapgpu_ctx *c)
{
    snapgpu_set_aligner_flags(c, o.stop_on_first_hit ? 1 : 0, o.explore_popular_seeds ? 1 : 0);      // -f, -x (single-end launches only)
    if (o.paired) { int rc = snapgpu_enable_paired(c, &o.pp); if (rc != SNAPGPU_OK) { fprintf(stderr, "snapgpu-sam: snapgpu_enable_paired failed (%d): %s\n", rc, snapgpu_last_error(c)); exit(1); } }
    if (o.om >= 0) {
        snapgpu_secondary_params sp; memset(&sp, 0, sizeof(sp));
        sp.max_edit_distance = o.om; sp.max_per_contig = o.mpc; sp.max_results = o.omax; sp.adjust_alignments = o.ae ? 1u : 0u;
        int rc = snapgpu_enable_secondary(c, &sp);
        if (rc != SNAPGPU_OK) { fprintf(stderr, "snapgpu-sam: snapgpu_enable_secondary failed (%d): %s\n", rc, snapgpu_last_error(c)); exit(1); }
    }
}
static snapgpu_ctx *ctx_for(const Options &o, FeederCtx &f, uint32_t max_len)
{
    int k = 0;
    while (k < 2 && max_len > class_len(o, k)) k++;
    while (k < 2 && class_len(o, k) == class_len(o, k + 1) && f.c[k] == NULL) k++;     // (classes that coincide: use the one that exists)
    if (f.c[k]) return f.c[k];
    for (int j = 0; j < 3; j++) if (f.c[j] && class_len(o, j) == class_len(o, k)) return f.c[k] = f.c[j];
    snapgpu_params p = o.p; p.max_read_len = class_len(o, k);
    int rc = snapgpu_create_replica_with_params(f.owner, f.device, 1, &p, &f.c[k]);
    if (rc != SNAPGPU_OK) { fprintf(stderr, "snapgpu-sam: snapgpu_create_replica_with_params failed (%d): %s\n", rc, snapgpu_last_error(f.owner)); exit(1); }
    configure_ctx(o, f.c[k]);
    return f.c[k];
}

// A cigar that does not fit ops_stride comes back as n_ops = -1 with nm = -2 (the reference's "cigarBuf too small" never happens: its
// buffer is large): call again with a larger stride rather than print a wrong record.
template <class F> static void with_growing_stride(Work &w, size_t n_rec, F call)
{
    for (;;) {
        w.ops.assign(n_rec * (size_t)w.ops_stride, 0);
        call();
        bool too_small = false;
        for (size_t i = 0; i < n_rec; i++) if (w.nm[i] == -2) { too_small = true; break; }
        if (!too_small) return;
        if (w.ops_stride >= 4096) die("a cigar needs more than 4096 operations");
        w.ops_stride *= 4;
    }
}

// Read::clip and the useless-read filter for a batch, and the reads to align gathered into one buffer: host work that does not need the GPU,
// done by the parser threads where there are several (the feeder threads' time belongs to the device)
static void prepare_single(const Options &o, Work &w)
{
    const Batch &b = w.b;
    const size_t n = b.n();
    w.front_clip.assign(n, 0); w.data_len.assign(n, 0); w.skip.assign(n, 1); w.to_align.clear(); w.ab.clear(); w.aq.clear(); w.ao.assign(1, 0);
    const bool fused = o.om < 0 && !o.ae;               // gpu_single's one-call path works on the batch itself: nothing to gather
    if (!fused) { w.ab.reserve(b.bases.size()); w.aq.reserve(b.quals.size()); w.ao.reserve(n + 1); }
    w.to_align.reserve(n);
    w.max_len = 0;
    for (size_t i = 0; i < n; i++) {
        const uint32_t U = (uint32_t)(b.offsets[i + 1] - b.offsets[i]);
        if (U > w.max_len) w.max_len = U;
        if (clip_read(o, b, i, w.front_clip[i], w.data_len[i])) {
            w.to_align.push_back((uint32_t)i); w.skip[i] = 0;
            if (fused) continue;
            const char *q = b.quals.data() + b.offsets[i] + w.front_clip[i], *s = b.bases.data() + b.offsets[i] + w.front_clip[i];
            w.ab.insert(w.ab.end(), s, s + w.data_len[i]); w.aq.insert(w.aq.end(), q, q + w.data_len[i]); w.ao.push_back(w.ab.size());
        }
    }
    w.prepared = true;
}

static bool g_index_has_alt = false;        // the index has ALT contigs: only then can a first-ALT result (an extra record) exist
static void gpu_single(const Options &o, FeederCtx &fc, Work &w)
{
    auto t_stage = std::chrono::steady_clock::now();
    auto lap = [&](std::atomic<unsigned long long> &acc) { const auto t = std::chrono::steady_clock::now(); acc += (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(t - t_stage).count(); t_stage = t; };
    const Batch &b = w.b;
    const size_t n = b.n();
    if (!w.prepared) prepare_single(o, w);            // (the parser threads of the mapped reader have done it already)
    snapgpu_ctx *ctx = ctx_for(o, fc, w.max_len);
    // ---- the common case in ONE call (snapgpu_align_sam_single): no secondary results, no -ae; one record per read unless a first-ALT result turns up
    std::vector<snapgpu_single_result> fused_res, fused_alt;
    bool fused_done = false;
    if (o.om < 0 && !o.ae && n > 0) {
        const bool want_alt = o.p.alt_awareness && g_index_has_alt;
        if (want_alt) { fused_res.resize(n); fused_alt.resize(n); }
        w.rec_read.resize(n); for (size_t i = 0; i < n; i++) w.rec_read[i] = (uint32_t)i;
        w.rec_secondary.assign(n, 0);
        w.flag.assign(n, 0); w.contig.assign(n, 0); w.mapq.assign(n, 0); w.n_ops.assign(n, 0); w.nm.assign(n, 0); w.pos.assign(n, 0);
        std::vector<int32_t> stale(n);
        w.ops_stride = o.ops_stride;
        lap(g_ns_prep);
        with_growing_stride(w, n, [&] {
            int rc2 = snapgpu_align_sam_single(ctx, (uint32_t)n, b.bases.data(), b.quals.data(), b.offsets.data(), w.front_clip.data(), w.data_len.data(), w.skip.data(),
                                               o.use_m ? 1 : 0, want_alt ? fused_res.data() : NULL, want_alt ? fused_alt.data() : NULL,
                                               w.flag.data(), w.contig.data(), w.pos.data(), w.mapq.data(), w.ops.data(), w.ops_stride, w.n_ops.data(), w.nm.data(), stale.data());
            if (rc2 != SNAPGPU_OK) fail_rc(ctx, "snapgpu_align_sam_single", rc2);
        });
        lap(g_ns_align);
        bool any_alt = false;
        if (want_alt) for (size_t i = 0; i < n && !any_alt; i++) any_alt = !w.skip[i] && fused_alt[i].status != SNAPGPU_NotFound;
        if (!any_alt) return;
        fused_done = true;                             // (rare: ALT records to add -- the records are built below from the results, as after the two-call path)
    }
    std::vector<int32_t> &front_clip = w.front_clip, &data_len = w.data_len;
    std::vector<uint32_t> &to_align = w.to_align;
    std::vector<char> &ab = w.ab, &aq = w.aq; std::vector<uint64_t> &ao = w.ao;
    std::vector<snapgpu_single_result> results(n), aligned_res(to_align.size()), alt_res(to_align.size());
    for (size_t i = 0; i < n; i++) {                                       // SingleAligner.cpp:215-225
        memset(&results[i], 0, sizeof(results[i]));
        results[i].status = SNAPGPU_NotFound; results[i].location = SNAPGPU_InvalidGenomeLocation32; results[i].score = -1;
    }
    // records to write: every read's primary, then its secondary results in the aligner's order (SingleAligner.cpp:300-318 -> writeReads)
    std::vector<snapgpu_single_result> rec_res;
    w.rec_read.clear(); w.rec_secondary.clear();
    int rc;
    if (fused_done) {                                  // the alignment is done: results per read, straight from the one call
        for (size_t i = 0; i < n; i++) {
            if (!w.skip[i]) results[i] = fused_res[i];
            w.rec_read.push_back((uint32_t)i); w.rec_secondary.push_back(0); rec_res.push_back(results[i]);
            if (!w.skip[i] && fused_alt[i].status != SNAPGPU_NotFound) { w.rec_read.push_back((uint32_t)i); w.rec_secondary.push_back(1); rec_res.push_back(fused_alt[i]); }
        }
    } else if (!to_align.empty()) {
        std::vector<snapgpu_single_result> sec; std::vector<uint32_t> nsec(to_align.size(), 0);
        uint32_t stride = 0;
        lap(g_ns_prep);
        if (o.om >= 0) {
            stride = 8;
            for (;;) {
                sec.assign((size_t)to_align.size() * stride, snapgpu_single_result());
                rc = snapgpu_align_single_secondary(ctx, (uint32_t)to_align.size(), ab.data(), aq.data(), ao.data(), aligned_res.data(), alt_res.data(),
                                                    sec.data(), stride, nsec.data());
                if (rc != SNAPGPU_W_SECONDARY_TRUNCATED) break;
                uint32_t need = stride; for (uint32_t v : nsec) if (v > need) need = v;
                stride = need;
            }
            // -ae with -om: the adjuster ran inside the call (finalizeSecondaryResults, BaseAligner.cpp:2444-2463) on the primary AND on
            // every secondary result, with the limitation described below for the primary: refuse the same reads here
            for (size_t a = 0; o.ae && rc == SNAPGPU_OK && a < to_align.size(); a++) {
                const size_t i = to_align[a];
                if (front_clip[i] == 0 && (uint64_t)data_len[i] == b.offsets[i + 1] - b.offsets[i]) continue;      // the reader clipped nothing
                for (uint32_t j = 0; j <= (nsec[a] < stride ? nsec[a] : stride); j++) {
                    const snapgpu_single_result &r = j == 0 ? aligned_res[a] : sec[a * (size_t)stride + (j - 1)];
                    if (r.status == SNAPGPU_NotFound) continue;
                    const int c = h_contig_at(r.location);
                    if (c >= 0 && r.location + data_len[i] + (long long)o.p.max_k + 2 > h_contig_end(c) - (long long)g_padding)
                        die("-ae: a quality-clipped read hangs over the end of its contig, which the adjuster does not reproduce (run with -C--)");
                }
            }
        } else {
            rc = snapgpu_align_single(ctx, (uint32_t)to_align.size(), ab.data(), aq.data(), ao.data(), aligned_res.data(), alt_res.data());
            if (rc == SNAPGPU_OK && o.ae) {                 // finalizeSecondaryResults has only the primary to adjust (BaseAligner.cpp:2444-2452)
                std::vector<int32_t> lens(to_align.size());
                for (size_t k = 0; k < to_align.size(); k++) lens[k] = (int32_t)(ao[k + 1] - ao[k]);
                rc = snapgpu_adjust_alignments(ctx, (uint32_t)to_align.size(), ab.data(), (uint64_t)ab.size(), ao.data(), lens.data(), aligned_res.data());
                // The adjuster is restated for reads the reader has not clipped (adjust.h): the reference settles a contig-end overhang on the
                // UNCLIPPED buffer (AlignmentAdjuster.cpp:167), which only differs for a clipped read that hangs over the end of its contig.
                // Such a read is refused rather than answered differently (-C-- switches the reader's clipping off).
                for (size_t a = 0; rc == SNAPGPU_OK && a < to_align.size(); a++) {
                    const size_t i = to_align[a];
                    if (front_clip[i] == 0 && (uint64_t)data_len[i] == b.offsets[i + 1] - b.offsets[i]) continue;
                    const snapgpu_single_result &r = aligned_res[a];
                    if (r.status == SNAPGPU_NotFound) continue;
                    const int c = h_contig_at(r.location);
                    if (c >= 0 && r.location + data_len[i] + (long long)o.p.max_k + 2 > h_contig_end(c) - (long long)g_padding)
                        die("-ae: a quality-clipped read hangs over the end of its contig, which the adjuster does not reproduce (run with -C--)");
                }
            }
        }
        if (rc != SNAPGPU_OK) fail_rc(ctx, "alignment", rc);
        lap(g_ns_align);
        std::vector<uint32_t> slot(n, 0xffffffffu);
        for (size_t k = 0; k < to_align.size(); k++) { results[to_align[k]] = aligned_res[k]; slot[to_align[k]] = (uint32_t)k; }
        for (size_t i = 0; i < n; i++) {
            w.rec_read.push_back((uint32_t)i); w.rec_secondary.push_back(0); rec_res.push_back(results[i]);
            if (o.om >= 0 && slot[i] != 0xffffffffu)
                for (uint32_t j = 0; j < nsec[slot[i]]; j++) { w.rec_read.push_back((uint32_t)i); w.rec_secondary.push_back(1); rec_res.push_back(sec[(size_t)slot[i] * stride + j]); }
            if (o.p.alt_awareness && slot[i] != 0xffffffffu && alt_res[slot[i]].status != SNAPGPU_NotFound) {      // writeReads(&firstALTResult, 1, firstIsPrimary = false)
                w.rec_read.push_back((uint32_t)i); w.rec_secondary.push_back(1); rec_res.push_back(alt_res[slot[i]]);
            }
        }
    } else for (size_t i = 0; i < n; i++) { w.rec_read.push_back((uint32_t)i); w.rec_secondary.push_back(0); rec_res.push_back(results[i]); }
    const size_t nr = w.rec_read.size();
    // the record batch: a read with secondary results appears once per record
    std::vector<char> rb, rq; std::vector<uint64_t> ro(1, 0); std::vector<int32_t> rfc(nr), rdl(nr);
    const bool one_to_one = nr == n;
    if (!one_to_one) {
        for (size_t r = 0; r < nr; r++) {
            const size_t i = w.rec_read[r];
            rb.insert(rb.end(), b.bases.begin() + (long)b.offsets[i], b.bases.begin() + (long)b.offsets[i + 1]);
            rq.insert(rq.end(), b.quals.begin() + (long)b.offsets[i], b.quals.begin() + (long)b.offsets[i + 1]);
            ro.push_back(rb.size()); rfc[r] = front_clip[i]; rdl[r] = data_len[i];
        }
    }
    w.flag.assign(nr, 0); w.contig.assign(nr, 0); w.mapq.assign(nr, 0); w.n_ops.assign(nr, 0); w.nm.assign(nr, 0); w.pos.assign(nr, 0);
    std::vector<int32_t> stale(nr);
    w.ops_stride = o.ops_stride;
    lap(g_ns_mid);
    with_growing_stride(w, nr, [&] {
        int rc2 = one_to_one
            ? snapgpu_sam_fields_single(ctx, (uint32_t)nr, b.bases.data(), b.quals.data(), b.offsets.data(), front_clip.data(), data_len.data(), rec_res.data(),
                                        o.use_m ? 1 : 0, w.flag.data(), w.contig.data(), w.pos.data(), w.mapq.data(), w.ops.data(), w.ops_stride, w.n_ops.data(),
                                        w.nm.data(), stale.data())
            : snapgpu_sam_fields_single(ctx, (uint32_t)nr, rb.data(), rq.data(), ro.data(), rfc.data(), rdl.data(), rec_res.data(), o.use_m ? 1 : 0,
                                        w.flag.data(), w.contig.data(), w.pos.data(), w.mapq.data(), w.ops.data(), w.ops_stride, w.n_ops.data(), w.nm.data(), stale.data());
        if (rc2 != SNAPGPU_OK) fail_rc(ctx, "snapgpu_sam_fields_single", rc2);
    });
    lap(g_ns_samcall);
    for (size_t r = 0; r < nr; r++) if (w.rec_secondary[r]) w.flag[r] |= 0x100;                  // SAM_SECONDARY (createSAMLine, SAM.cpp:1477-1479)
}

// Several batches in ONE GPU call.  The host stages want fine grain -- a batch is parsed by one thread and formatted by one thread, so the
// pipeline fills and drains in the time ONE batch takes -- and the GPU wants ~1 M reads per launch (a launch ends on its slowest read).
// So the batches stay small (131 072 reads) and a feeder hands `-g` of them to snapgpu_align_sam_single as one batch: inputs gathered into
// the feeder's own buffers, outputs scattered back to the batches they belong to, which then go on to the formatters separately.
// Anything but the one-call path (secondary results, -ae, a first-ALT record in the group) is done batch by batch as before.
// The group's buffers are the feeder's own and live as long as it does, so they are worth page-locking: the library takes any host pointer,
// and a pageable one goes through the runtime's staging buffers at a fraction of the link's rate -- ~0.4 GB per call of 1 M reads, up and
// down.  The tool talks to the library through the C ABI only; the HIP runtime the library brought into the process is looked up by name
// (none in the emulator's build of the tool: the buffers then stay pageable).  SNAPGPU_SAM_PIN=0 turns it off.
struct PinnedRange {
    void *p = nullptr; size_t bytes = 0;
    typedef int (*reg_fn)(void *, size_t, unsigned); typedef int (*unreg_fn)(void *);
    static reg_fn reg() { static reg_fn f = (getenv("SNAPGPU_SAM_PIN") && atoi(getenv("SNAPGPU_SAM_PIN")) == 0) ? nullptr : (reg_fn)dlsym(RTLD_DEFAULT, "hipHostRegister"); return f; }
    static unreg_fn unreg() { static unreg_fn f = (unreg_fn)dlsym(RTLD_DEFAULT, "hipHostUnregister"); return f; }
    void release() { if (p && unreg()) (void)unreg()(p); p = nullptr; bytes = 0; }
    // the vector's storage as it is now (after the call's resizes): registered again only when it moved or grew
    template <typename T> void cover(std::vector<T> &v) {
        void *q = (void *)v.data(); const size_t b = v.capacity() * sizeof(T);
        if (q == p && b == bytes) return;
        release();
        static const size_t min_bytes = getenv("SNAPGPU_SAM_PIN_MIN") ? (size_t)atoll(getenv("SNAPGPU_SAM_PIN_MIN")) : ((size_t)1 << 20);
        if (!reg() || !unreg() || b < min_bytes || b == 0) return;
        if (reg()(q, b, 1u /* hipHostRegisterPortable */) == 0) { p = q; bytes = b; }
    }
    ~PinnedRange() { release(); }
};
struct GroupBuf {
    std::vector<char> b, q; std::vector<uint64_t> off; std::vector<int32_t> fc, dl, flag, contig, mapq, n_ops, nm, stale; std::vector<uint8_t> skip;
    std::vector<int64_t> pos; std::vector<uint32_t> ops; std::vector<snapgpu_single_result> res, alt;
    PinnedRange pin[16];
    // resize without ever letting a vector move while its storage is registered: growth releases the registration first (and takes headroom)
    template <typename T> static void fit(std::vector<T> &v, size_t n, PinnedRange &r) { if (n > v.capacity()) { r.release(); v.reserve(n + n / 4); } v.resize(n); }
    void pin_all() {
        pin[0].cover(b); pin[1].cover(q); pin[2].cover(off); pin[3].cover(fc); pin[4].cover(dl); pin[5].cover(flag); pin[6].cover(contig); pin[7].cover(mapq);
        pin[8].cover(n_ops); pin[9].cover(nm); pin[10].cover(stale); pin[11].cover(skip); pin[12].cover(pos); pin[13].cover(ops); pin[14].cover(res); pin[15].cover(alt);
    }
    ~GroupBuf() { for (auto &r : pin) r.release(); }          // (before the vectors go)
};
static void gpu_single_group(const Options &o, FeederCtx &fc, std::vector<Work *> &ws, GroupBuf &g)
{
    const bool fusable = o.om < 0 && !o.ae;
    if (ws.size() == 1 || !fusable) { for (Work *w : ws) gpu_single(o, fc, *w); return; }
    auto t_stage = std::chrono::steady_clock::now();
    auto lap = [&](std::atomic<unsigned long long> &acc) { const auto t = std::chrono::steady_clock::now(); acc += (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(t - t_stage).count(); t_stage = t; };
    size_t n = 0, nbytes = 0; uint32_t max_len = 0;
    for (Work *w : ws) { if (!w->prepared) prepare_single(o, *w); n += w->b.n(); nbytes += w->b.bases.size(); if (w->max_len > max_len) max_len = w->max_len; }
    if (n == 0) { for (Work *w : ws) gpu_single(o, fc, *w); return; }
    snapgpu_ctx *ctx = ctx_for(o, fc, max_len);
    GroupBuf::fit(g.b, nbytes, g.pin[0]); GroupBuf::fit(g.q, nbytes, g.pin[1]); GroupBuf::fit(g.off, n + 1, g.pin[2]); GroupBuf::fit(g.fc, n, g.pin[3]);
    GroupBuf::fit(g.dl, n, g.pin[4]); GroupBuf::fit(g.skip, n, g.pin[11]);
    size_t at = 0, ab = 0;
    for (Work *w : ws) {
        const Batch &b = w->b; const size_t m = b.n();
        memcpy(g.b.data() + ab, b.bases.data(), b.bases.size()); memcpy(g.q.data() + ab, b.quals.data(), b.quals.size());
        for (size_t i = 0; i < m; i++) g.off[at + i] = b.offsets[i] + ab;
        memcpy(g.fc.data() + at, w->front_clip.data(), m * 4); memcpy(g.dl.data() + at, w->data_len.data(), m * 4); memcpy(g.skip.data() + at, w->skip.data(), m);
        at += m; ab += b.bases.size();
    }
    g.off[n] = ab;
    const bool want_alt = o.p.alt_awareness && g_index_has_alt;
    if (want_alt) { GroupBuf::fit(g.res, n, g.pin[14]); GroupBuf::fit(g.alt, n, g.pin[15]); }
    GroupBuf::fit(g.flag, n, g.pin[5]); GroupBuf::fit(g.contig, n, g.pin[6]); GroupBuf::fit(g.mapq, n, g.pin[7]); GroupBuf::fit(g.n_ops, n, g.pin[8]);
    GroupBuf::fit(g.nm, n, g.pin[9]); GroupBuf::fit(g.stale, n, g.pin[10]); GroupBuf::fit(g.pos, n, g.pin[12]);
    uint32_t stride = o.ops_stride;
    lap(g_ns_prep);
    for (;;) {
        GroupBuf::fit(g.ops, 0, g.pin[13]); GroupBuf::fit(g.ops, n * (size_t)stride, g.pin[13]);      // (zero-filled: resize from empty)
        g.pin_all();
        const int rc = snapgpu_align_sam_single(ctx, (uint32_t)n, g.b.data(), g.q.data(), g.off.data(), g.fc.data(), g.dl.data(), g.skip.data(), o.use_m ? 1 : 0,
                                                want_alt ? g.res.data() : NULL, want_alt ? g.alt.data() : NULL,
                                                g.flag.data(), g.contig.data(), g.pos.data(), g.mapq.data(), g.ops.data(), stride, g.n_ops.data(), g.nm.data(), g.stale.data());
        if (rc != SNAPGPU_OK) fail_rc(ctx, "snapgpu_align_sam_single", rc);
        bool too_small = false;
        for (size_t i = 0; i < n && !too_small; i++) too_small = g.nm[i] == -2;
        if (!too_small) break;
        if (stride >= 4096) die("a cigar needs more than 4096 operations");
        stride *= 4;
    }
    lap(g_ns_align);
    if (want_alt) for (size_t i = 0; i < n; i++) if (!g.skip[i] && g.alt[i].status != SNAPGPU_NotFound) { for (Work *w : ws) gpu_single(o, fc, *w); return; }   // (rare: ALT records: batch by batch)
    at = 0;
    for (Work *w : ws) {
        const size_t m = w->b.n();
        w->rec_read.resize(m); for (size_t i = 0; i < m; i++) w->rec_read[i] = (uint32_t)i;
        w->rec_secondary.assign(m, 0);
        w->flag.assign(g.flag.begin() + (long)at, g.flag.begin() + (long)(at + m)); w->contig.assign(g.contig.begin() + (long)at, g.contig.begin() + (long)(at + m));
        w->mapq.assign(g.mapq.begin() + (long)at, g.mapq.begin() + (long)(at + m)); w->n_ops.assign(g.n_ops.begin() + (long)at, g.n_ops.begin() + (long)(at + m));
        w->nm.assign(g.nm.begin() + (long)at, g.nm.begin() + (long)(at + m)); w->pos.assign(g.pos.begin() + (long)at, g.pos.begin() + (long)(at + m));
        w->ops_stride = stride; w->ops.assign(g.ops.begin() + (long)(at * stride), g.ops.begin() + (long)((at + m) * stride));
        at += m;
    }
    lap(g_ns_mid);
}

static void gpu_paired(const Options &o, FeederCtx &fc, Work &w)
{
    const Batch &b = w.b;
    const size_t n = b.n(), np = n / 2;
    w.max_len = 0;
    for (size_t i = 0; i < n; i++) { const uint32_t U = (uint32_t)(b.offsets[i + 1] - b.offsets[i]); if (U > w.max_len) w.max_len = U; }
    snapgpu_ctx *ctx = ctx_for(o, fc, w.max_len);
    std::vector<int32_t> front_clip(n, 0), data_len(n, 0);
    std::vector<char> useful(n, 0);
    for (size_t i = 0; i < n; i++) useful[i] = clip_read(o, b, i, front_clip[i], data_len[i]);
    std::vector<uint32_t> to_align;                                     // pairs with at least one useful mate (PairedAligner.cpp:680-682)
    std::vector<char> ab, aq; std::vector<uint64_t> ao(1, 0);
    for (size_t k = 0; k < np; k++) {
        if (!useful[2 * k] && !useful[2 * k + 1]) continue;
        to_align.push_back((uint32_t)k);
        for (size_t i = 2 * k; i < 2 * k + 2; i++) {
            const char *q = b.quals.data() + b.offsets[i], *s = b.bases.data() + b.offsets[i];
            ab.insert(ab.end(), s + front_clip[i], s + front_clip[i] + data_len[i]); aq.insert(aq.end(), q + front_clip[i], q + front_clip[i] + data_len[i]); ao.push_back(ab.size());
        }
    }
    std::vector<snapgpu_paired_result> results(np), ares(to_align.size()), aalt(to_align.size());
    for (size_t k = 0; k < np; k++) {
        memset(&results[k], 0, sizeof(results[k]));
        for (int v = 0; v < 2; v++) { results[k].status[v] = SNAPGPU_NotFound; results[k].location[v] = SNAPGPU_InvalidGenomeLocation32; results[k].score[v] = -1; }
    }
    // secondary results (-om): paired ones and each mate's single-end ones (PairedAligner.cpp:727-779)
    std::vector<snapgpu_paired_result> sec; std::vector<uint32_t> nsec(to_align.size(), 0), nssec(2 * to_align.size(), 0);
    std::vector<snapgpu_single_result> ssec;
    uint32_t sec_stride = 0, ssec_stride = 0;
    if (!to_align.empty()) {
        int rc;
        if (o.om >= 0) {
            sec_stride = 8; ssec_stride = 16;
            for (;;) {
                sec.assign(to_align.size() * (size_t)sec_stride, snapgpu_paired_result()); ssec.assign(to_align.size() * (size_t)ssec_stride, snapgpu_single_result());
                rc = snapgpu_align_paired_secondary(ctx, (uint32_t)to_align.size(), ab.data(), aq.data(), ao.data(), ares.data(), aalt.data(),
                                                    sec.data(), sec_stride, nsec.data(), ssec.data(), ssec_stride, nssec.data());
                if (rc != SNAPGPU_W_SECONDARY_TRUNCATED) break;
                for (size_t k = 0; k < to_align.size(); k++) {
                    if (nsec[k] > sec_stride) sec_stride = nsec[k];
                    if (nssec[2 * k] + nssec[2 * k + 1] > ssec_stride) ssec_stride = nssec[2 * k] + nssec[2 * k + 1];
                }
            }
        } else rc = snapgpu_align_paired(ctx, (uint32_t)to_align.size(), ab.data(), aq.data(), ao.data(), ares.data(), aalt.data());
        if (rc != SNAPGPU_OK) fail_rc(ctx, "snapgpu_align_paired", rc);
        for (size_t k = 0; k < to_align.size(); k++) results[to_align[k]] = ares[k];
    }
    // what gets written, in the reference's order (PairedAligner.cpp:874-880 -> SimpleReadWriter::writePairs): the pair's results (primary first),
    // read 0's single-end secondary results, read 1's, then -- with -ea -- the first ALT result as a pair of its own
    std::vector<snapgpu_paired_result> pu_res; std::vector<snapgpu_single_result> su_res;
    w.emit.clear(); w.pu_pair.clear(); w.pu_secondary.clear(); w.su_read.clear();
    {
        std::vector<uint32_t> slot(np, 0xffffffffu);
        for (size_t k = 0; k < to_align.size(); k++) slot[to_align[k]] = (uint32_t)k;
        auto add_pair_unit = [&](size_t k, const snapgpu_paired_result &r, bool secondary) {
            w.emit.push_back(Work::Emit{(uint32_t)pu_res.size(), 0}); w.pu_pair.push_back((uint32_t)k); w.pu_secondary.push_back(secondary ? 1 : 0); pu_res.push_back(r);
        };
        for (size_t k = 0; k < np; k++) {
            add_pair_unit(k, results[k], false);
            const uint32_t a = slot[k];
            if (a == 0xffffffffu) continue;
            if (o.om >= 0) {
                for (uint32_t j = 0; j < nsec[a]; j++) add_pair_unit(k, sec[(size_t)a * sec_stride + j], true);
                uint32_t at = 0;
                for (int v = 0; v < 2; v++) {
                    // (the reference applies the FIRST single result's clippingForReadAdjustment to every one of the mate's single records:
                    //  `singleResults[whichRead]->clippingForReadAdjustment`, ReadWriter.cpp:516)
                    const int32_t first_adj = nssec[2 * a + v] ? ssec[(size_t)a * ssec_stride + at].clipping_for_read_adjustment : 0;
                    for (uint32_t j = 0; j < nssec[2 * a + v]; j++, at++) {
                        snapgpu_single_result r = ssec[(size_t)a * ssec_stride + at];
                        r.clipping_for_read_adjustment = first_adj;
                        w.emit.push_back(Work::Emit{(uint32_t)su_res.size(), 1}); w.su_read.push_back((uint32_t)(2 * k + (size_t)v)); su_res.push_back(r);
                    }
                }
            }
            if (o.p.emit_alt_alignments && (aalt[a].status[0] != SNAPGPU_NotFound || aalt[a].status[1] != SNAPGPU_NotFound)) add_pair_unit(k, aalt[a], false);
        }
    }
    const size_t nu = pu_res.size(), nrec = 2 * nu, ns = su_res.size();
    w.flag.assign(nrec, 0); w.contig.assign(nrec, 0); w.mapq.assign(nrec, 0); w.n_ops.assign(nrec, 0); w.nm.assign(nrec, 0); w.rnext.assign(nrec, 0);
    w.first_written.assign(nu, 0); w.pos.assign(nrec, 0); w.pnext.assign(nrec, 0); w.tlen.assign(nrec, 0);
    w.s_flag.assign(ns, 0); w.s_contig.assign(ns, 0); w.s_mapq.assign(ns, 0); w.s_n_ops.assign(ns, 0); w.s_nm.assign(ns, 0); w.s_pos.assign(ns, 0);
    w.ops_stride = w.s_ops_stride = o.ops_stride;
    if (nu == np && ns == 0) {                                          // one result per pair: one call over the batch as it is
        std::vector<int32_t> stale(nrec);
        with_growing_stride(w, nrec, [&] {
            int rc = snapgpu_sam_fields_paired(ctx, (uint32_t)nu, b.bases.data(), b.quals.data(), b.offsets.data(), front_clip.data(), data_len.data(), pu_res.data(),
                                               o.use_m ? 1 : 0, w.flag.data(), w.contig.data(), w.pos.data(), w.mapq.data(), w.ops.data(), w.ops_stride, w.n_ops.data(),
                                               w.nm.data(), w.rnext.data(), w.pnext.data(), w.tlen.data(), w.first_written.data(), stale.data());
            if (rc != SNAPGPU_OK) fail_rc(ctx, "snapgpu_sam_fields_paired", rc);
        });
        return;
    }
    // Several results per pair.  The reference writes them one after the other through the SAME two Read objects, and one piece of their state
    // survives from record to record: the additional BACK clipping that the affine-gap writer sets when it turns a leading insertion of a
    // reverse-complement alignment into a soft clip (SAM.cpp:1676-1680 / ReadWriter.cpp:545-551; setAdditionalFrontClipping is re-set for every
    // result, setAdditionalBackClipping never is: Read.h:537-553).  So the j-th entries of all pairs are computed together (round j), and what a
    // record leaves behind -- readable off its leading soft clip -- shortens the data length the next entry of the same read is given.
    std::vector<std::vector<Work::Emit>> per_pair(np);
    for (const Work::Emit &e : w.emit) per_pair[e.single ? w.su_read[e.unit] / 2 : w.pu_pair[e.unit]].push_back(e);
    size_t n_rounds = 0;
    for (size_t k = 0; k < np; k++) if (per_pair[k].size() > n_rounds) n_rounds = per_pair[k].size();
    const bool host_all = getenv("SNAPGPU_SAM_HOST_LOOP") != NULL;      // (tests: every record of a multi-result batch through the host loop)
    for (;;) {                                                           // (again with a larger stride when a cigar did not fit)
        bool too_small = false;
        std::vector<int32_t> carry(n, 0);
        w.ops.assign(nrec * (size_t)w.ops_stride, 0); w.s_ops.assign(ns * (size_t)w.s_ops_stride, 0);
        auto ag_branch = [&](int used_ag, int score) { return o.p.use_affine_gap && (used_ag != 0 || score > 0); };
        // the state a written record leaves on its Read: `passed_len` is the data length the record was computed with
        // (`hr` / `paired_rec`: the result the record was computed from, for the two cases in which the record alone does not say what it left
        //  behind -- a record given up after clipping adjustments, and a reverse-complement record at the first base of a contig, whose leading
        //  soft clip may have been cut short there: the record loop is then run on the host for that one record, with the Read's state as it
        //  is, and ITS back clipping is what the next record of the read starts from.  Until round 4 these two cases stopped the program.)
        auto leave_behind = [&](size_t rd, bool more, bool ag, int status_in, int flag, int64_t pos, int n_ops, const uint32_t *ops, int clipped_before, int passed_len,
                                const HostRes &hr, bool paired_rec) {
            if (!more || !ag) return;
            const size_t U = (size_t)(b.offsets[rd + 1] - b.offsets[rd]);
            if (((flag & 0x4) && status_in != SNAPGPU_NotFound) || ((flag & 0x10) && n_ops > 0 && pos == 1)) {
                carry[rd] = host_record(o, ctx, b.bases.data() + b.offsets[rd], b.quals.data() + b.offsets[rd], (int)U, front_clip[rd], data_len[rd], carry[rd], hr, paired_rec).back_after;
                if (getenv("SNAPGPU_SAM_VERBOSE")) fprintf(stderr, "snapgpu-sam: one record's state recomputed on the host (%s), back clipping left: %d\n", (flag & 0x4) ? "given up after clipping adjustments" : "reverse complement at the first base of a contig", carry[rd]);
                return;
            }
            if (!(flag & 0x10) || n_ops <= 0) return;
            const int lead = (ops[0] & 15u) == 4u ? (int)(ops[0] >> 4) : 0;
            const int left = lead - ((int)U - passed_len - front_clip[rd]) - clipped_before;
            if (left < 0) die("internal error: leading soft clip shorter than the read's own clipping");
            if (left > 0) carry[rd] = left;                             // (device-computed records start from carry == 0: the others go through host_record)
        };
        for (size_t j = 0; j < n_rounds; j++) {
            std::vector<uint32_t> pus, sus, pus_host, sus_host;          // this round's pair units / single units; `_host`: a read arrives with back clipping
            for (size_t k = 0; k < np; k++) {
                if (j >= per_pair[k].size()) continue;
                const Work::Emit &e = per_pair[k][j];
                if (e.single) (host_all || carry[w.su_read[e.unit]] > 0 ? sus_host : sus).push_back(e.unit);
                else (host_all || carry[2 * k] > 0 || carry[2 * k + 1] > 0 ? pus_host : pus).push_back(e.unit);
            }
            for (uint32_t u : pus_host) {                                 // the record loop on the host, with the Reads' state as it is (exact)
                const size_t k = w.pu_pair[u];
                const snapgpu_paired_result &r = pu_res[u];
                HostRec rec[2];
                for (size_t v = 0; v < 2; v++) {
                    const size_t i = 2 * k + v;
                    const HostRes hr = {r.status[v], r.direction[v], r.score[v], r.mapq[v], r.clipping_for_read_adjustment[v], r.used_affine_gap_scoring[v],
                                        r.bases_clipped_before[v], r.bases_clipped_after[v], r.supplementary[v], (long long)r.location[v]};
                    rec[v] = host_record(o, ctx, b.bases.data() + b.offsets[i], b.quals.data() + b.offsets[i], (int)(b.offsets[i + 1] - b.offsets[i]), front_clip[i], data_len[i], carry[i], hr, true);
                    carry[i] = rec[v].back_after;
                }
                for (size_t v = 0; v < 2; v++) {
                    const size_t dst = 2 * u + v;
                    int fl, ct, rn; long long ps, pn, tl;
                    host_mate_info(rec[v], rec[1 - v], v == 0, r.aligned_as_pair != 0, fl, ct, ps, rn, pn, tl);
                    if ((uint32_t)(rec[v].n_ops > 0 ? rec[v].n_ops : 0) > w.ops_stride) { too_small = true; continue; }
                    w.flag[dst] = fl | (w.pu_secondary[u] ? 0x100 : 0); w.contig[dst] = ct; w.pos[dst] = ps; w.mapq[dst] = rec[v].mapq; w.n_ops[dst] = rec[v].n_ops; w.nm[dst] = rec[v].nm;
                    w.rnext[dst] = rn; w.pnext[dst] = pn; w.tlen[dst] = tl;
                    for (int c2 = 0; c2 < rec[v].n_ops; c2++) w.ops[dst * (size_t)w.ops_stride + (size_t)c2] = rec[v].ops[(size_t)c2];
                }
                const unsigned long long l0 = rec[0].final_loc < 0 ? ~0ull : (unsigned long long)rec[0].final_loc, l1 = rec[1].final_loc < 0 ? ~0ull : (unsigned long long)rec[1].final_loc;
                w.first_written[u] = l0 <= l1 ? 0 : 1;                    // ReadWriter.cpp:481-488
            }
            for (uint32_t r : sus_host) {
                const size_t rd = w.su_read[r];
                const snapgpu_single_result &sr = su_res[r];
                const HostRes hr = {sr.status, sr.direction, sr.score, sr.mapq, sr.clipping_for_read_adjustment, sr.used_affine_gap_scoring, sr.bases_clipped_before, sr.bases_clipped_after,
                                    sr.supplementary, (long long)sr.location};
                const HostRec rec = host_record(o, ctx, b.bases.data() + b.offsets[rd], b.quals.data() + b.offsets[rd], (int)(b.offsets[rd + 1] - b.offsets[rd]), front_clip[rd], data_len[rd], carry[rd], hr, false);
                carry[rd] = rec.back_after;
                if ((uint32_t)(rec.n_ops > 0 ? rec.n_ops : 0) > w.s_ops_stride) { too_small = true; continue; }
                w.s_flag[r] = rec.flag | 0x100; w.s_contig[r] = rec.contig; w.s_pos[r] = rec.pos; w.s_mapq[r] = rec.mapq; w.s_n_ops[r] = rec.n_ops; w.s_nm[r] = rec.nm;
                for (int c2 = 0; c2 < rec.n_ops; c2++) w.s_ops[r * (size_t)w.s_ops_stride + (size_t)c2] = rec.ops[(size_t)c2];
            }
            if (!pus.empty()) {
                const size_t m = pus.size();
                std::vector<char> rb, rq; std::vector<uint64_t> ro(1, 0); std::vector<int32_t> rfc(2 * m), rdl(2 * m), stale(2 * m);
                std::vector<snapgpu_paired_result> rr(m);
                for (size_t x = 0; x < m; x++) {
                    rr[x] = pu_res[pus[x]];
                    for (size_t v = 0; v < 2; v++) {
                        const size_t i = 2 * (size_t)w.pu_pair[pus[x]] + v;
                        rb.insert(rb.end(), b.bases.begin() + (long)b.offsets[i], b.bases.begin() + (long)b.offsets[i + 1]);
                        rq.insert(rq.end(), b.quals.begin() + (long)b.offsets[i], b.quals.begin() + (long)b.offsets[i + 1]);
                        ro.push_back(rb.size()); rfc[2 * x + v] = front_clip[i]; rdl[2 * x + v] = data_len[i] - carry[i];
                    }
                }
                std::vector<int32_t> f(2 * m), c(2 * m), mq(2 * m), no(2 * m), nm(2 * m), rn(2 * m), fw(m); std::vector<int64_t> ps(2 * m), pn(2 * m), tl(2 * m);
                std::vector<uint32_t> ops(2 * m * (size_t)w.ops_stride);
                int rc = snapgpu_sam_fields_paired(ctx, (uint32_t)m, rb.data(), rq.data(), ro.data(), rfc.data(), rdl.data(), rr.data(), o.use_m ? 1 : 0,
                                                   f.data(), c.data(), ps.data(), mq.data(), ops.data(), w.ops_stride, no.data(), nm.data(), rn.data(), pn.data(), tl.data(), fw.data(), stale.data());
                if (rc != SNAPGPU_OK) fail_rc(ctx, "snapgpu_sam_fields_paired", rc);
                for (size_t x = 0; x < m; x++) {
                    const size_t u = pus[x], k = w.pu_pair[u];
                    w.first_written[u] = fw[x];
                    for (size_t v = 0; v < 2; v++) {
                        const size_t src = 2 * x + v, dst = 2 * u + v;
                        if (nm[src] == -2) too_small = true;
                        w.flag[dst] = f[src] | (w.pu_secondary[u] ? 0x100 : 0); w.contig[dst] = c[src]; w.pos[dst] = ps[src]; w.mapq[dst] = mq[src]; w.n_ops[dst] = no[src];
                        w.nm[dst] = nm[src]; w.rnext[dst] = rn[src]; w.pnext[dst] = pn[src]; w.tlen[dst] = tl[src];
                        memcpy(&w.ops[dst * (size_t)w.ops_stride], &ops[src * (size_t)w.ops_stride], (size_t)w.ops_stride * 4);
                        const HostRes hr = {rr[x].status[v], rr[x].direction[v], rr[x].score[v], rr[x].mapq[v], rr[x].clipping_for_read_adjustment[v], rr[x].used_affine_gap_scoring[v],
                                            rr[x].bases_clipped_before[v], rr[x].bases_clipped_after[v], rr[x].supplementary[v], (long long)rr[x].location[v]};
                        leave_behind(2 * k + v, j + 1 < per_pair[k].size(), ag_branch(rr[x].used_affine_gap_scoring[v], rr[x].score[v]), rr[x].status[v], f[src], ps[src], no[src],
                                     &ops[src * (size_t)w.ops_stride], rr[x].bases_clipped_before[v], rdl[src], hr, true);
                    }
                }
            }
            if (!sus.empty()) {                                          // records of their own, without mate information (format->writeRead, ReadWriter.cpp:521-527)
                const size_t m = sus.size();
                std::vector<char> sb, sq; std::vector<uint64_t> so(1, 0); std::vector<int32_t> sfc(m), sdl(m), stale(m);
                std::vector<snapgpu_single_result> rr(m);
                for (size_t x = 0; x < m; x++) {
                    const size_t i = w.su_read[sus[x]];
                    rr[x] = su_res[sus[x]];
                    sb.insert(sb.end(), b.bases.begin() + (long)b.offsets[i], b.bases.begin() + (long)b.offsets[i + 1]);
                    sq.insert(sq.end(), b.quals.begin() + (long)b.offsets[i], b.quals.begin() + (long)b.offsets[i + 1]);
                    so.push_back(sb.size()); sfc[x] = front_clip[i]; sdl[x] = data_len[i] - carry[i];
                }
                std::vector<int32_t> f(m), c(m), mq(m), no(m), nm(m); std::vector<int64_t> ps(m);
                std::vector<uint32_t> ops(m * (size_t)w.s_ops_stride);
                int rc = snapgpu_sam_fields_single(ctx, (uint32_t)m, sb.data(), sq.data(), so.data(), sfc.data(), sdl.data(), rr.data(), o.use_m ? 1 : 0,
                                                   f.data(), c.data(), ps.data(), mq.data(), ops.data(), w.s_ops_stride, no.data(), nm.data(), stale.data());
                if (rc != SNAPGPU_OK) fail_rc(ctx, "snapgpu_sam_fields_single", rc);
                for (size_t x = 0; x < m; x++) {
                    const size_t r = sus[x], rd = w.su_read[r], k = rd / 2;
                    if (nm[x] == -2) too_small = true;
                    w.s_flag[r] = f[x] | 0x100; w.s_contig[r] = c[x]; w.s_pos[r] = ps[x]; w.s_mapq[r] = mq[x]; w.s_n_ops[r] = no[x]; w.s_nm[r] = nm[x];
                    memcpy(&w.s_ops[r * (size_t)w.s_ops_stride], &ops[x * (size_t)w.s_ops_stride], (size_t)w.s_ops_stride * 4);
                    // (a single-end record is written without the aligner's soft clipping unless it went through the affine-gap writer: sam_fields.h)
                    const bool ag = ag_branch(rr[x].used_affine_gap_scoring, rr[x].score);
                    const HostRes hr = {rr[x].status, rr[x].direction, rr[x].score, rr[x].mapq, rr[x].clipping_for_read_adjustment, rr[x].used_affine_gap_scoring, rr[x].bases_clipped_before,
                                        rr[x].bases_clipped_after, rr[x].supplementary, (long long)rr[x].location};
                    leave_behind(rd, j + 1 < per_pair[k].size(), ag, rr[x].status, f[x], ps[x], no[x], &ops[x * (size_t)w.s_ops_stride], ag ? rr[x].bases_clipped_before : 0, sdl[x], hr, false);
                }
            }
        }
        if (!too_small) break;
        if (w.ops_stride >= 4096) die("a cigar needs more than 4096 operations");
        w.ops_stride *= 4; w.s_ops_stride = w.ops_stride;
    }
}

// ---------------------------------------------------------------------------------------- formatting stage
static void put_bam_record(std::string &o, const std::vector<Contig> &contigs, const char *qname, size_t qn, int flag, int contig, long long pos1,
                           int mapq, const uint32_t *ops, int n_ops, int mate_contig, long long pnext1, long long tlen,
                           const char *s, const char *q, size_t U, int nm, bool with_qs, int qs);
static inline void put_cigar(std::string &o, const Work &w, size_t i)
{
    if (w.n_ops[i] < 0) { o.push_back('*'); return; }
    for (int c = 0; c < w.n_ops[i]; c++) { const uint32_t op = w.ops[i * (size_t)w.ops_stride + (size_t)c]; put_uint(o, op >> 4); o.push_back("MIDNSHP=X"[op & 15]); }
}
static inline void put_seq_qual(std::string &o, const char *s, const char *q, size_t U, bool rc)
{
    const size_t at = o.size();
    o.resize(at + 2 * U + 1);
    char *d = &o[at];
    if (rc) { for (size_t j = 0; j < U; j++) { d[U - 1 - j] = complement(s[j]); d[U + 1 + U - 1 - j] = q[j]; } }
    else { memcpy(d, s, U); memcpy(d + U + 1, q, U); }
    d[U] = '\t';
}
static const char AUX_TAIL[] = "\tRG:Z:FASTQ\tPL:Z:Illumina\tPU:Z:pu\tLB:Z:lb\tSM:Z:sm";

static void format_single(const std::vector<Contig> &contigs, Work &w)
{
    const Batch &b = w.b;
    std::string &o = w.text;
    o.clear(); o.reserve(w.rec_read.size() * 420);
    for (size_t i = 0; i < w.rec_read.size(); i++) {
        const size_t rd = w.rec_read[i];
        const char *s = b.bases.data() + b.offsets[rd], *q = b.quals.data() + b.offsets[rd];
        const size_t U = (size_t)(b.offsets[rd + 1] - b.offsets[rd]);
        const char *nm = b.names.data() + b.name_off[rd]; size_t nl = b.name_off[rd + 1] - b.name_off[rd];
        const void *sp = memchr(nm, ' ', nl);                              // "illegal in SAM: truncate at the space" (SAM.cpp:2001-2004)
        if (sp) nl = (size_t)((const char *)sp - nm);
        if (w.bam) {
            put_bam_record(o, contigs, nm, nl, w.flag[i], w.contig[i], w.pos[i], w.mapq[i], w.ops.data() + i * (size_t)w.ops_stride, w.n_ops[i], -1, 0, 0, s, q, U, w.nm[i], false, 0);
            w.mapped += (w.flag[i] & 0x4) == 0;
            continue;
        }
        o.append(nm, nl); o.push_back('\t'); put_int(o, w.flag[i]); o.push_back('\t');
        if (w.contig[i] >= 0) o += contigs[(size_t)w.contig[i]].name; else o.push_back('*');
        o.push_back('\t'); put_int(o, w.pos[i]); o.push_back('\t'); put_int(o, w.mapq[i]); o.push_back('\t');
        put_cigar(o, w, i);
        o += "\t*\t0\t0\t";
        put_seq_qual(o, s, q, U, (w.flag[i] & 0x10) != 0);
        o += "\tPG:Z:SNAP\tNM:i:"; put_int(o, w.nm[i]); o += AUX_TAIL; o.push_back('\n');
        w.mapped += (w.flag[i] & 0x4) == 0;
    }
}

static void format_paired(const std::vector<Contig> &contigs, Work &w)
{
    const Batch &b = w.b;
    std::string &o = w.text;
    o.clear(); o.reserve(w.emit.size() * 880);
    for (const Work::Emit &e : w.emit) {
        if (e.single) {
            // a mate's single-end secondary result: SAMFormat::writeRead without a mate -- the whole read id (the "/1" stays: the call is given
            // getIdLength(), ReadWriter.cpp:521), no mate fields, no QS
            const size_t r = e.unit, rd = w.su_read[r];
            const char *s = b.bases.data() + b.offsets[rd], *q = b.quals.data() + b.offsets[rd];
            const size_t U = (size_t)(b.offsets[rd + 1] - b.offsets[rd]);
            const char *nm = b.names.data() + b.name_off[rd]; size_t nl = b.name_off[rd + 1] - b.name_off[rd];
            const void *sp = memchr(nm, ' ', nl);
            if (sp) nl = (size_t)((const char *)sp - nm);
            if (w.bam) {
                put_bam_record(o, contigs, nm, nl, w.s_flag[r], w.s_contig[r], w.s_pos[r], w.s_mapq[r], w.s_ops.data() + r * (size_t)w.s_ops_stride, w.s_n_ops[r], -1, 0, 0, s, q, U, w.s_nm[r], false, 0);
            } else {
                o.append(nm, nl); o.push_back('\t'); put_int(o, w.s_flag[r]); o.push_back('\t');
                if (w.s_contig[r] >= 0) o += contigs[(size_t)w.s_contig[r]].name; else o.push_back('*');
                o.push_back('\t'); put_int(o, w.s_pos[r]); o.push_back('\t'); put_int(o, w.s_mapq[r]); o.push_back('\t');
                if (w.s_n_ops[r] < 0) o.push_back('*');
                else for (int c = 0; c < w.s_n_ops[r]; c++) { const uint32_t op = w.s_ops[r * (size_t)w.s_ops_stride + (size_t)c]; put_uint(o, op >> 4); o.push_back("MIDNSHP=X"[op & 15]); }
                o += "\t*\t0\t0\t";
                put_seq_qual(o, s, q, U, (w.s_flag[r] & 0x10) != 0);
                o += "\tPG:Z:SNAP\tNM:i:"; put_int(o, w.s_nm[r]); o += AUX_TAIL; o.push_back('\n');
            }
            w.mapped += (w.s_flag[r] & 0x4) == 0;
            continue;
        }
        const size_t u = e.unit, k = w.pu_pair[u];
        // QNAME: the /1 /2 suffixes go when both names carry them (ReadWriter.cpp:392-404)
        const char *n0 = b.names.data() + b.name_off[2 * k], *n1 = b.names.data() + b.name_off[2 * k + 1];
        size_t idl[2] = { (size_t)(b.name_off[2 * k + 1] - b.name_off[2 * k]), (size_t)(b.name_off[2 * k + 2] - b.name_off[2 * k + 1]) };
        if (idl[0] == idl[1] && idl[0] > 2 && n0[idl[0] - 2] == '/' && n1[idl[0] - 2] == '/') {
            const char c0 = n0[idl[0] - 1], c1 = n1[idl[1] - 1];
            if ((c0 == '1' || c0 == '2') && (c1 == '1' || c1 == '2') && c0 != c1) { idl[0] -= 2; idl[1] -= 2; }
        }
        for (int ord = 0; ord < 2; ord++) {
            const int v = ord == 0 ? w.first_written[u] : 1 - w.first_written[u];
            const size_t i = 2 * u + (size_t)v;                                             // fields
            const size_t rd = 2 * k + (size_t)v, rm = 2 * k + (size_t)(1 - v);              // the read and its mate in the batch
            const char *s = b.bases.data() + b.offsets[rd], *q = b.quals.data() + b.offsets[rd];
            const size_t U = (size_t)(b.offsets[rd + 1] - b.offsets[rd]);
            const char *nm = v == 0 ? n0 : n1;
            size_t qn = idl[v];
            const void *sp = memchr(nm, ' ', qn);
            if (sp) qn = (size_t)((const char *)sp - nm);
            int mqs = 0;                                                // QS: the mate's qualities >= 15, summed (SAM.cpp:1826-1837)
            { const unsigned char *mq = (const unsigned char *)b.quals.data() + b.offsets[rm]; const size_t mu = (size_t)(b.offsets[rm + 1] - b.offsets[rm]);
              for (size_t j = 0; j < mu; j++) { const int x = (int)mq[j] - '!'; mqs += x >= 15 ? (x != 255) * x : 0; } }
            if (w.bam) {
                const int mate_contig = w.rnext[i] == -2 ? w.contig[i] : w.rnext[i];
                put_bam_record(o, contigs, nm, qn, w.flag[i], w.contig[i], w.pos[i], w.mapq[i], w.ops.data() + i * (size_t)w.ops_stride, w.n_ops[i],
                               mate_contig, w.pnext[i], w.tlen[i], s, q, U, w.nm[i], true, mqs);
                w.mapped += (w.flag[i] & 0x4) == 0;
                continue;
            }
            o.append(nm, qn); o.push_back('\t'); put_int(o, w.flag[i]); o.push_back('\t');
            if (w.contig[i] >= 0) o += contigs[(size_t)w.contig[i]].name; else o.push_back('*');
            o.push_back('\t'); put_int(o, w.pos[i]); o.push_back('\t'); put_int(o, w.mapq[i]); o.push_back('\t');
            put_cigar(o, w, i); o.push_back('\t');
            if (w.rnext[i] == -2) o.push_back('='); else if (w.rnext[i] >= 0) o += contigs[(size_t)w.rnext[i]].name; else o.push_back('*');
            o.push_back('\t'); put_int(o, w.pnext[i]); o.push_back('\t'); put_int(o, (int)w.tlen[i]); o.push_back('\t');
            put_seq_qual(o, s, q, U, (w.flag[i] & 0x10) != 0);
            o += "\tPG:Z:SNAP\tNM:i:"; put_int(o, w.nm[i]); o += AUX_TAIL; o += "\tQS:i:"; put_int(o, mqs); o.push_back('\n');
            w.mapped += (w.flag[i] & 0x4) == 0;
        }
    }
}

// ---------------------------------------------------------------------------------------- BAM (SNAPLib/Bam.cpp)
// One record as BAMFormat::writeRead / writePairs lay it out (Bam.cpp:1312-1508, 1033-1309): the same computed fields as the SAM line, binary.
//   refID / next_refID are ORIGINAL contig numbers (the header lists the contigs in the FASTA's order, Bam.cpp:1003-1021),
//   bin = reg2bin over the reference span of the cigar (an unmapped read: its mate's position, else -1; Bam.cpp:1473-1477),
//   aux: the default read group's fields (RG PL PU LB SM, Z), PG:Z:SNAP, NM:C (one byte: -1 reads as 255), and QS:i for mates (Bam.cpp:1510-1650).
static int bam_reg2bin(int beg, int end)                                   // Bam.cpp:523-535
{
    --end;
    if (beg >> 14 == end >> 14) return ((1 << 15) - 1) / 7 + (beg >> 14);
    if (beg >> 17 == end >> 17) return ((1 << 12) - 1) / 7 + (beg >> 17);
    if (beg >> 20 == end >> 20) return ((1 << 9) - 1) / 7 + (beg >> 20);
    if (beg >> 23 == end >> 23) return ((1 << 6) - 1) / 7 + (beg >> 23);
    if (beg >> 26 == end >> 26) return ((1 << 3) - 1) / 7 + (beg >> 26);
    return 0;
}
static inline void put_le32(std::string &o, uint32_t v) { char c[4] = {(char)(v & 0xff), (char)((v >> 8) & 0xff), (char)((v >> 16) & 0xff), (char)(v >> 24)}; o.append(c, 4); }
static const char BAM_RG_AUX[] = "RGZFASTQ\0PLZIllumina\0PUZpu\0LBZlb\0SMZsm";            // (+ the terminating NUL of the literal)
struct BamSeqCode { uint8_t c[256]; BamSeqCode() { memset(c, 0xf, 256); const char *t = "=ACMGRSVTWYHKDBN"; for (int i = 1; i < 16; i++) c[(unsigned char)t[i]] = (uint8_t)i; } };
static const BamSeqCode BAM_SEQ;                                             // Bam.cpp:505-510: anything unknown is N

static void put_bam_record(std::string &o, const std::vector<Contig> &contigs, const char *qname, size_t qn, int flag, int contig, long long pos1,
                           int mapq, const uint32_t *ops, int n_ops, int mate_contig, long long pnext1, long long tlen,
                           const char *s, const char *q, size_t U, int nm, bool with_qs, int qs)
{
    if (qn > 254) die("BAM format: QNAME must be shorter than 255 characters");
    const bool rc = (flag & 0x10) != 0;
    const int n_cig = n_ops > 0 ? n_ops : 0;
    static const int ref_base[16] = {1, 0, 1, 1, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};       // Bam.cpp:270
    int ref_len = n_cig > 0 ? 0 : (int)U;
    for (int c = 0; c < n_cig; c++) ref_len += ref_base[ops[c] & 0xf] * (int)(ops[c] >> 4);
    int bin;
    if (!(flag & 0x4)) bin = bam_reg2bin((int)pos1 - 1, (int)pos1 - 1 + ref_len);
    else if (with_qs && !(flag & 0x8)) bin = bam_reg2bin((int)pnext1 - 1, (int)pnext1);
    else bin = bam_reg2bin(-1, 0);
    const size_t aux_len = sizeof(BAM_RG_AUX) + 8 + 4 + (with_qs ? 7 : 0);
    const size_t block = 32 + (qn + 1) + 4 * (size_t)n_cig + (U + 1) / 2 + U + aux_len;     // without the block_size word itself
    const size_t at = o.size();
    o.reserve(at + 4 + block);
    put_le32(o, (uint32_t)block);
    put_le32(o, (uint32_t)(contig >= 0 ? contigs[(size_t)contig].orig : -1));
    put_le32(o, (uint32_t)((int)pos1 - 1));
    put_le32(o, ((uint32_t)bin << 16) | ((uint32_t)(mapq & 0xff) << 8) | (uint32_t)(qn + 1));
    put_le32(o, ((uint32_t)flag << 16) | (uint32_t)n_cig);
    put_le32(o, (uint32_t)U);
    put_le32(o, (uint32_t)(mate_contig >= 0 ? contigs[(size_t)mate_contig].orig : -1));
    put_le32(o, (uint32_t)((int)pnext1 - 1));
    put_le32(o, (uint32_t)(int)(tlen >= 0 ? (tlen & 0x7fffffff) : -((-tlen) & 0x7fffffff)));
    o.append(qname, qn); o.push_back('\0');
    for (int c = 0; c < n_cig; c++) put_le32(o, ops[c]);
    {   // SEQ (4 bits per base) and QUAL (Phred), in the record's orientation
        const size_t a0 = o.size();
        o.resize(a0 + (U + 1) / 2 + U);
        uint8_t *sq = (uint8_t *)&o[a0], *ql = sq + (U + 1) / 2;
        for (size_t j = 0; j < U; j++) {
            const char base = rc ? complement(s[U - 1 - j]) : s[j];
            const uint8_t code = BAM_SEQ.c[(unsigned char)base];
            if (j & 1) sq[j >> 1] |= code; else sq[j >> 1] = (uint8_t)(code << 4);
            ql[j] = (uint8_t)((rc ? q[U - 1 - j] : q[j]) - '!');
        }
    }
    o.append(BAM_RG_AUX, sizeof(BAM_RG_AUX));
    o.append("PGZSNAP", 8);
    o.append("NMC", 3); o.push_back((char)(uint8_t)nm);
    if (with_qs) { o.append("QSi", 3); put_le32(o, (uint32_t)qs); }
    if (o.size() - at != 4 + block) die("internal error: BAM record size");
}

// BGZF (SAM/BAM specification, section 4.1): gzip members of at most 64 KiB with the block size in an extra field; concatenated members
// are a valid gzip file, which is what lets every formatter thread compress its own batch.
static void bgzf_append(std::string &out, const char *data, size_t n)
{
    const size_t CHUNK = 0xff00;
    for (size_t at = 0; at < n || (n == 0 && at == 0); at += CHUNK) {
        const size_t len = n - at < CHUNK ? n - at : CHUNK;
        z_stream zs; memset(&zs, 0, sizeof(zs));
        if (deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) die("deflateInit2 failed");
        std::vector<unsigned char> buf(len + len / 8 + 64);
        zs.next_in = (Bytef *)(data + at); zs.avail_in = (uInt)len; zs.next_out = buf.data(); zs.avail_out = (uInt)buf.size();
        if (deflate(&zs, Z_FINISH) != Z_STREAM_END) die("deflate failed");
        const size_t clen = buf.size() - zs.avail_out;
        deflateEnd(&zs);
        const size_t bsize = 18 + clen + 8;
        if (bsize > 0x10000) die("internal error: BGZF block too large");
        const unsigned char hdr[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, (unsigned char)((bsize - 1) & 0xff), (unsigned char)((bsize - 1) >> 8)};
        out.append((const char *)hdr, 18);
        out.append((const char *)buf.data(), clen);
        put_le32(out, (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef *)(data + at), (uInt)len));
        put_le32(out, (uint32_t)len);
        if (n == 0) break;
    }
}

// ---------------------------------------------------------------------------------------- main
int main(int argc, char **argv)
{
    Options o;
    o.paired = argc >= 2 && strcmp(argv[1], "paired") == 0;
    if (argc < (o.paired ? 5 : 4) || (!o.paired && strcmp(argv[1], "single") != 0))
        die("usage: snapgpu-sam single <index-dir> <reads.fq> -o <out.sam|out.bam> | paired <index-dir> <r1.fq> <r2.fq> -o <out.sam|out.bam>  [-d N] [-G-] [-=] [-M] [-om N] [-omax N] [-mpc N] [-ae] [-ea] [-mrl N] [-b N] [-gpus N] [-q N] [-t N]");
    const std::string index_dir = argv[2], fastq = argv[3], fastq2 = o.paired ? argv[4] : "";
    std::string out_path;
    snapgpu_default_params(&o.p);
    o.p.max_read_len = 400;                                                // per-wave buffers; SNAPGPU_MAX_READ_LEN raises it (<= 1000), as for snap-aligner-gpu
    if (const char *e = getenv("SNAPGPU_MAX_READ_LEN")) { const int v = atoi(e); if (v >= 50 && v <= 1000) o.p.max_read_len = (uint32_t)v; }
    std::string cl = argv[1];
    for (int i = 2; i < argc; i++) { cl += " "; cl += argv[i]; }
    for (int i = o.paired ? 5 : 4; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "-o" && i + 1 < argc) out_path = argv[++i];
        else if (a == "-d" && i + 1 < argc) o.p.max_k = (uint32_t)atoi(argv[++i]);
        else if (a == "-G-") o.p.use_affine_gap = 0;
        else if (a == "-=") o.use_m = false;
        else if (a == "-M") o.use_m = true;
        else if (a.size() == 4 && a.compare(0, 2, "-C") == 0 && strchr("+-", a[2]) && strchr("+-", a[3])) { o.clip_front = a[2] == '+'; o.clip_back = a[3] == '+'; }   // AlignerOptions.cpp: -Cxx
        else if (a == "-om" && i + 1 < argc) o.om = atoi(argv[++i]);                       // secondary alignments (AlignerOptions.cpp:70-72)
        else if (a == "-omax" && i + 1 < argc) o.omax = atoll(argv[++i]);
        else if (a == "-mpc" && i + 1 < argc) o.mpc = atoi(argv[++i]);
        else if (a == "-ae") o.ae = true;                                                    // AlignerOptions.cpp:476
        else if (a == "-f") o.stop_on_first_hit = true;                                      // AlignerOptions.cpp:573
        else if (a == "-x") o.explore_popular_seeds = true;                                  // AlignerOptions.cpp:570
        else if (a == "-ea") o.p.emit_alt_alignments = 1;                                  // the first ALT alignment as an extra record (SingleAligner.cpp:320-322)
        else if (a == "-D" && i + 1 < argc) o.p.extra_search_depth = (uint32_t)atoi(argv[++i]);
        else if (a == "-mrl" && i + 1 < argc) o.min_read_len = (unsigned)atoi(argv[++i]);
        else if (a == "-b" && i + 1 < argc) o.batch_reads = (size_t)atoll(argv[++i]);
        else if (a == "-g" && i + 1 < argc) o.group = (size_t)atoll(argv[++i]);
        else if (a == "-gpus" && i + 1 < argc) o.n_gpus = atoi(argv[++i]);
        else if (a == "-q" && i + 1 < argc) o.ctx_per_gpu = atoi(argv[++i]);
        else if (a == "-t" && i + 1 < argc) o.n_format = atoi(argv[++i]);  // host threads that format records (the reference's -t counts aligner threads)
        else if (a == "-tp" && i + 1 < argc) o.n_parse = atoi(argv[++i]);  // host threads that parse a plain FASTQ file
        else if (a == "-seqread") o.seqread = true;
        else if (a == "-passes" && i + 1 < argc) o.passes = atoi(argv[++i]);   // (measurement: the whole FASTQ -> SAM stream N times over the resident index; the output is the last pass's)
        else die("option not supported: ", a.c_str());
    }
    if (out_path.empty()) die("-o <out.sam | out.bam> is required");
    if (o.ae && o.paired) die("-ae is implemented for `single` only (the paired-end aligners' use of AlignmentAdjuster is not: DESIGN.md section 16)");
    if (o.ae && o.clip_front) die("-ae with front clipping (-C+x) is not supported: the adjuster is restated for reads the reader has not clipped at the front");
    o.bam = out_path.size() > 4 && out_path.compare(out_path.size() - 4, 4, ".bam") == 0;     // by extension, like the reference (AlignerOptions.cpp)
    const bool batch_auto = o.batch_reads == 0;
    if (batch_auto) o.batch_reads = 262144;                                // (the mapped reader, which knows the file's size, picks below)
    if (o.batch_reads < (o.paired ? 2u : 1u)) die("-b must be at least 1 (2 for paired)");
    if (o.paired) o.batch_reads &= ~(size_t)1;
    // feeders per GPU: a launch lasts as long as its slowest read (~150 ms when the batch holds one of the heavy ones), so batches of 131 072
    // reads need several launches in flight to keep the chip busy: 20 M reads go through at 2.24 / 2.47 / 2.50 M reads/s with 3 / 6 / 8
    // feeders (profiles/r04zy).  The paired-end launches are long already.
    // Round 5: the GPU gets ~1 M single-end reads per call (gpu_single_group: bench.py's launch size -- the tail of a launch then costs 10 %,
    // not half of it) through ONE call (snapgpu_align_sam_single) and the 192-position kernel variant: three feeders cover the
    // copies of one batch with the kernels of the others (20 M reads: 4.78 / 4.56 / 4.74 M reads/s with 3 / 4 / 6 feeders, profiles/r05g).
    if (o.ctx_per_gpu == 0) o.ctx_per_gpu = 3;
    if (o.ctx_per_gpu < 1 || o.ctx_per_gpu > 8) die("-q must be in [1, 8]");
    // cigar ops per record: about 2 * edits + soft clips; grown on demand when a record needs more (with_growing_stride)
    { uint32_t need = 2 * (o.p.max_k + o.p.extra_search_depth) + 8; o.ops_stride = (need + 3u) & ~3u; if (o.ops_stride < 16) o.ops_stride = 16; }

    const auto t_process = std::chrono::steady_clock::now();
    std::vector<Contig> contigs; uint64_t n_bases = 0; uint32_t padding = 0;
    load_contigs(index_dir, contigs, n_bases, padding);
    g_contigs = &contigs; g_n_bases = n_bases; g_padding = padding;

    // ---- contexts: GPU 0 reads the index, the other GPUs get it over RCCL, every further feeder on a GPU shares that GPU's blobs
    int visible = snapgpu_device_count();
    if (visible <= 0) { snapgpu_ctx *none = NULL; int rc = snapgpu_create_from_directory(index_dir.c_str(), &o.p, 0, &none); fail_rc(none, "snapgpu_create_from_directory", rc); }
    if (o.n_gpus <= 0 || o.n_gpus > visible) o.n_gpus = visible;
    std::vector<snapgpu_ctx *> primary((size_t)o.n_gpus, NULL);
    snapgpu_default_paired_params(&o.pp);
    o.pp.min_read_length = o.min_read_len;
    for (const Contig &c : contigs) if (c.is_alt) g_index_has_alt = true;
    // every context starts in read-length class 0 (FeederCtx): the index owner too
    snapgpu_params p0 = o.p; p0.max_read_len = class_len(o, 0);
    int rc = snapgpu_create_from_directory(index_dir.c_str(), &p0, 0, &primary[0]);
    if (rc != SNAPGPU_OK) fail_rc(primary[0], "snapgpu_create_from_directory", rc);
    for (int g = 1; g < o.n_gpus; g++) {
        rc = snapgpu_create_replica(primary[0], g, 0, &primary[(size_t)g]);
        if (rc != SNAPGPU_OK) fail_rc(primary[0], "snapgpu_create_replica", rc);
    }
    if (o.n_gpus > 1) {
        rc = snapgpu_broadcast_index(primary.data(), o.n_gpus);
        if (rc != SNAPGPU_OK) fail_rc(primary[0], "snapgpu_broadcast_index", rc);
    }
    std::vector<FeederCtx> fctx;                                            // one per feeder thread
    for (int g = 0; g < o.n_gpus; g++) {
        for (int k = 0; k < o.ctx_per_gpu; k++) {
            FeederCtx f; f.owner = primary[(size_t)g]; f.device = g;
            if (k == 0) f.c[0] = primary[(size_t)g];
            else {
                rc = snapgpu_create_replica(primary[(size_t)g], g, 1, &f.c[0]);
                if (rc != SNAPGPU_OK) fail_rc(primary[(size_t)g], "snapgpu_create_replica", rc);
            }
            configure_ctx(o, f.c[0]);
            fctx.push_back(f);
        }
    }
    // host threads: formatting a record is ~1 us of text work, so one GPU's ~5 M records/s want a few dozen formatters (the cap of 16 of
    // rounds 2-3 was the pipeline's bottleneck on a 256-thread host); parsers likewise
    { const unsigned hc = std::thread::hardware_concurrency() ? std::thread::hardware_concurrency() : 8;
      if (o.n_format <= 0) { unsigned v = hc / 3; o.n_format = (int)(v < 4 ? 4 : (v > 64 ? 64 : v)); }
      if (o.n_parse <= 0) { unsigned v = hc / 8; o.n_parse = (int)(v < 2 ? 2 : (v > 24 ? 24 : v)); } }

    if (o.passes < 1 || o.passes > 100) die("-passes must be in [1, 100]");
    unsigned long long total = 0, mapped = 0;
    bool use_map = false;
    auto t_ready = std::chrono::steady_clock::now();
    double s_load = 0.0;
    std::vector<double> pass_s;
    WorkPool pool(256);
    for (int pass = 0; pass < o.passes; pass++) {
    total = 0; mapped = 0;
    if (pass > 0) (void)unlink(out_path.c_str());        // (-passes: the previous pass's output goes BEFORE this pass is timed -- replacing a 5.6 GB file at the end of a pass
                                                         //  cost it 1.3 s of page-cache work that streaming a FASTQ file into a new SAM file does not have: profiles/r06j)
    const std::string partial_path = out_path + ".partial";
    FILE *out = fopen(partial_path.c_str(), "wb");
    if (!out) die("cannot create ", partial_path.c_str());
    g_partial_path = partial_path;
    setvbuf(out, NULL, _IOFBF, 8u << 20);
    // header (SAM.cpp:1232-1295); @SQ lines go by ORIGINAL contig number: the index builder moves ALT contigs behind the regular ones
    // (FASTA.cpp:359-384), the header keeps the FASTA's order (getContigByOriginalContigNumber, SAM.cpp:1291)
    // The BAM header is the same text inside the binary header, followed by the contig table in the same (original) order (Bam.cpp:970-1031).
    {
        std::string text = "@HD\tVN:1.6\tGO:query\n@RG\tID:FASTQ\tPL:Illumina\tPU:pu\tLB:lb\tSM:sm\n@PG\tID:SNAP\tPN:SNAP\tCL:" + cl + "\tVN:2.0.5\n";
        std::string refs;
        std::vector<size_t> by_orig(contigs.size());
        bool perm = true;
        std::vector<char> seen(contigs.size(), 0);
        for (size_t c = 0; c < contigs.size(); c++) { if (contigs[c].orig < 0 || (size_t)contigs[c].orig >= contigs.size() || seen[(size_t)contigs[c].orig]) { perm = false; break; } seen[(size_t)contigs[c].orig] = 1; by_orig[(size_t)contigs[c].orig] = c; }
        if (o.bam && !perm) die("the index's original contig numbers are not a permutation: cannot number the BAM reference sequences");
        for (size_t k = 0; k < contigs.size(); k++) {
            const size_t c = perm ? by_orig[k] : k;
            const uint64_t end = c + 1 < contigs.size() ? contigs[c + 1].begin : n_bases;
            const unsigned long long len = (unsigned long long)(end - contigs[c].begin - padding);
            text += "@SQ\tSN:" + contigs[c].name + "\tLN:" + std::to_string(len) + (contigs[c].is_alt ? "\tAH:*" : "") + "\n";
            put_le32(refs, (uint32_t)contigs[c].name.size() + 1); refs += contigs[c].name; refs.push_back('\0'); put_le32(refs, (uint32_t)len);
        }
        if (o.bam) {
            std::string h = "BAM\1";
            put_le32(h, (uint32_t)text.size()); h += text; put_le32(h, (uint32_t)contigs.size()); h += refs;
            std::string z; bgzf_append(z, h.data(), h.size());
            if (fwrite(z.data(), 1, z.size(), out) != z.size()) die("write error on ", out_path.c_str());
        } else if (fwrite(text.data(), 1, text.size(), out) != text.size()) die("write error on ", out_path.c_str());
    }

    t_ready = std::chrono::steady_clock::now();                             // the index is resident, the contexts exist: the streaming part starts here
    if (pass == 0) s_load = std::chrono::duration<double>(t_ready - t_process).count();
    // ---- the pipeline
    // single-end batches per GPU call: enough to make ~1 M reads, but never so many that the feeders of a small file have nothing to share
    size_t group = o.group ? o.group : (o.paired ? 1 : 8);
    if (group < 1) group = 1;
    if (group > 64) group = 64;
    Queue<Work *> q_parsed(fctx.size() * (o.paired ? 2 : group + 2) + 2), q_aligned((size_t)o.n_format * 2 + 2 + group * fctx.size());
    std::mutex done_m; std::condition_variable done_cv; std::map<uint64_t, Work *> done;
    std::atomic<uint64_t> n_batches(0); std::atomic<bool> reader_done(false);

    // ---- input: a plain FASTQ file is mapped, indexed by line count and parsed batch by batch by o.n_parse threads; gzip input, -seqread
    // and files whose lines do not come in fours go through the sequential reader
    MappedFastq mf, mf2;
    use_map = !o.seqread && mf.open(fastq.c_str()) && (!o.paired || mf2.open(fastq2.c_str()));
    if (use_map) {
        const auto t0 = std::chrono::steady_clock::now();
        use_map = mf.index(o.n_parse) && (!o.paired || mf2.index(o.n_parse));
        if (use_map && o.paired && mf.n_lines != mf2.n_lines) die(mf.n_lines > mf2.n_lines ? "the second FASTQ file has fewer reads than the first" : "the second FASTQ file has more reads than the first");
        if (use_map && getenv("SNAPGPU_SAM_VERBOSE")) fprintf(stderr, "snapgpu-sam: %llu FASTQ records indexed in %.2f s\n", (unsigned long long)(mf.n_lines / 4), std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    if (!use_map) { mf.close(); mf2.close(); }
    std::vector<std::thread> parsers;
    std::thread reader;
    if (use_map) {
        const uint64_t n_records = mf.n_lines / 4;
        if (batch_auto && o.paired) {       // enough batches to keep every feeder busy (four each), each as large as that allows: 64 K .. 512 K reads
            uint64_t v = 2 * n_records / (4 * (uint64_t)fctx.size());
            v = v < 65536 ? 65536 : (v > 524288 ? 524288 : v);
            o.batch_reads = (size_t)(v & ~(uint64_t)1);
        } else if (batch_auto) o.batch_reads = 131072;       // (single end: small batches for the host stages, grouped for the GPU: gpu_single_group)
        const uint64_t per_batch = o.paired ? o.batch_reads / 2 : o.batch_reads;           // records of EACH file per batch
        const uint64_t n_units = (n_records + per_batch - 1) / per_batch;
        n_batches = n_units; reader_done = true;
        // (a small file: no feeder takes more batches per call than leaves the others their share -- with every batch available at once the first feeder
        //  to wake would otherwise take eight and a file of a million reads would run on one feeder of one GPU)
        if (!o.group) { const size_t share = (size_t)(n_units / (uint64_t)fctx.size()); if (group > (share ? share : 1)) group = share ? share : 1; }
        std::shared_ptr<std::atomic<uint64_t>> next_unit = std::make_shared<std::atomic<uint64_t>>(0);
        std::shared_ptr<std::atomic<int>> parsers_left = std::make_shared<std::atomic<int>>(o.n_parse);
        for (int t = 0; t < o.n_parse; t++)
            parsers.emplace_back([&, next_unit, parsers_left, n_records, per_batch, n_units] {
                for (;;) {
                    const uint64_t u = next_unit->fetch_add(1);
                    if (u >= n_units) break;
                    const uint64_t r0 = u * per_batch, r1 = r0 + per_batch < n_records ? r0 + per_batch : n_records;
                    Work *w = pool.get();
                    StageTimer *pt = new StageTimer(g_ns_parse);
                    w->b.clear(); w->b.seq = u; w->bam = o.bam;
                    size_t at = mf.line_start(4 * r0), at2 = o.paired ? mf2.line_start(4 * r0) : 0;
                    for (uint64_t r = r0; r < r1; r++) {
                        at = parse_mapped_record(mf, at, w->b, o.p.max_read_len);
                        if (o.paired) at2 = parse_mapped_record(mf2, at2, w->b, o.p.max_read_len);
                    }
                    if (!o.paired) prepare_single(o, *w);
                    delete pt;
                    q_parsed.push(w);
                }
                if (--*parsers_left == 0) { q_parsed.close(); std::lock_guard<std::mutex> l(done_m); done_cv.notify_all(); }
            });
    } else
    reader = std::thread([&] {
        LineReader in, in2;
        in.open(fastq.c_str());
        if (o.paired) in2.open(fastq2.c_str());
        uint64_t seq = 0;
        bool eof = false;
        while (!eof) {
            Work *w = pool.get();
            w->b.clear(); w->b.seq = seq; w->bam = o.bam;
            while (w->b.n() < o.batch_reads) {
                if (!next_read(in, w->b, o.p.max_read_len)) { eof = true; break; }
                if (o.paired && !next_read(in2, w->b, o.p.max_read_len)) die("the second FASTQ file has fewer reads than the first");
            }
            if (eof && o.paired) { Batch probe; probe.clear(); if (next_read(in2, probe, o.p.max_read_len)) die("the second FASTQ file has more reads than the first"); }
            if (w->b.n() == 0) { pool.put(w); break; }
            seq++;
            q_parsed.push(w);
        }
        in.close(); in2.close();
        n_batches = seq; reader_done = true;
        q_parsed.close();
        { std::lock_guard<std::mutex> l(done_m); done_cv.notify_all(); }
    });
    std::vector<std::thread> feeders, formatters;
    std::atomic<int> feeders_left((int)fctx.size());
    for (size_t t = 0; t < fctx.size(); t++)
        feeders.emplace_back([&, t] {
            if (o.paired) {
                Work *w;
                while (q_parsed.pop(w)) { gpu_paired(o, fctx[t], *w); q_aligned.push(w); }
            } else {
                GroupBuf gb; std::vector<Work *> ws;
                for (;;) {
                    ws.clear();
                    { StageTimer st(g_ns_wait_in); if (q_parsed.pop_upto(ws, group, std::chrono::milliseconds(20)) == 0) break; }
                    gpu_single_group(o, fctx[t], ws, gb);
                    { StageTimer st(g_ns_wait_out); for (Work *w : ws) q_aligned.push(w); }
                }
            }
            if (--feeders_left == 0) q_aligned.close();
        });
    for (int t = 0; t < o.n_format; t++)
        formatters.emplace_back([&] {
            Work *w;
            while (q_aligned.pop(w)) {
                { StageTimer st(g_ns_format); if (o.paired) format_paired(contigs, *w); else format_single(contigs, *w); }
                if (o.bam) { std::string z; z.reserve(w->text.size() / 3 + 64); bgzf_append(z, w->text.data(), w->text.size()); w->text.swap(z); }
                std::lock_guard<std::mutex> l(done_m); done[w->b.seq] = w; done_cv.notify_all();
            }
        });
    const auto t_pipe = std::chrono::steady_clock::now();
    double t_first_s = -1.0;
    for (uint64_t next = 0;; next++) {                                      // the writer: batches in input order
        Work *w = NULL;
        {
            StageTimer st(g_ns_wait_writer);
            std::unique_lock<std::mutex> l(done_m);
            done_cv.wait(l, [&] { return done.count(next) || (reader_done && next >= n_batches); });
            if (!done.count(next)) break;
            w = done[next]; done.erase(next);
        }
        { StageTimer st(g_ns_write); if (fwrite(w->text.data(), 1, w->text.size(), out) != w->text.size()) die("write error on ", out_path.c_str()); }
        if (t_first_s < 0) t_first_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_ready).count();
        total += w->b.n(); mapped += w->mapped;
        pool.put(w);
    }
    const auto t_written = std::chrono::steady_clock::now();
    if (reader.joinable()) reader.join();
    for (auto &t : parsers) t.join();
    mf.close(); mf2.close();
    for (auto &t : feeders) t.join();
    for (auto &t : formatters) t.join();
    const auto t_joined = std::chrono::steady_clock::now();
    if (o.bam) { std::string eof; bgzf_append(eof, "", 0); if (fwrite(eof.data(), 1, eof.size(), out) != eof.size()) die("write error on ", out_path.c_str()); }     // the empty end-of-file block
    if (fclose(out) != 0) die("write error on ", out_path.c_str());
    if (rename(partial_path.c_str(), out_path.c_str()) != 0) die("cannot rename the finished output to ", out_path.c_str());
    g_partial_path.clear();
    pass_s.push_back(std::chrono::duration<double>(std::chrono::steady_clock::now() - t_ready).count());
    if (getenv("SNAPGPU_SAM_VERBOSE"))
        fprintf(stderr, "snapgpu-sam: pass %d timeline: pipeline up at %.3f s, first batch written at %.3f s, last at %.3f s, threads joined at %.3f s, file closed and renamed at %.3f s\n", pass + 1,
                std::chrono::duration<double>(t_pipe - t_ready).count(), t_first_s, std::chrono::duration<double>(t_written - t_ready).count(),
                std::chrono::duration<double>(t_joined - t_ready).count(), pass_s.back());
    if (o.passes > 1) fprintf(stderr, "snapgpu-sam: pass %d of %d: %llu reads in %.3f s = %.0f reads/s\n", pass + 1, o.passes, total, pass_s.back(), pass_s.back() > 0 ? (double)total / pass_s.back() : 0.0);
    }
    for (size_t t = fctx.size(); t-- > 0;) {                                // sharers before the owner of the blobs they share
        for (int k = 2; k >= 0; k--) {
            snapgpu_ctx *c = fctx[t].c[k];
            if (!c || c == fctx[t].owner) continue;
            bool seen = false; for (int j = 0; j < k; j++) seen = seen || fctx[t].c[j] == c;
            if (!seen) snapgpu_destroy(c);
        }
    }
    for (size_t g = primary.size(); g-- > 0;) snapgpu_destroy(primary[g]);
    fprintf(stderr, "snapgpu-sam: %llu reads, %llu mapped records, %d GPU(s) x %d feeder(s), %d formatter thread(s)\n", total, mapped, o.n_gpus, o.ctx_per_gpu, o.n_format);
    {   // (AlignerContext.cpp:489-543 prints reads/s over the alignment phase, the index load apart: the same split here)
        const double s_stream = pass_s.back();                              // (the last pass; with -passes N each pass printed its own line above)
        fprintf(stderr, "snapgpu-sam: index resident after %.2f s; FASTQ -> %s in %.2f s = %.0f reads/s (%s reader, %d parser thread(s))\n", s_load, o.bam ? "BAM" : "SAM", s_stream,
                s_stream > 0 ? (double)total / s_stream : 0.0, use_map ? "mapped" : "sequential", use_map ? o.n_parse : 1);
        if (getenv("SNAPGPU_SAM_VERBOSE")) {
            fprintf(stderr, "snapgpu-sam: thread-seconds: parse %.2f | feeders: prepare %.2f, align call %.2f, records %.2f, SAM-fields call %.2f | format %.2f | write %.2f\n",
                    g_ns_parse.load() * 1e-9, g_ns_prep.load() * 1e-9, g_ns_align.load() * 1e-9, g_ns_mid.load() * 1e-9, g_ns_samcall.load() * 1e-9, g_ns_format.load() * 1e-9, g_ns_write.load() * 1e-9);
            // the same per THREAD and per PASS: the wall time each stage would take alone with its threads (a stage near the pass's own time is the pipeline's limit)
            const double np = (double)o.passes, nf = (double)fctx.size();
            fprintf(stderr, "snapgpu-sam: waiting, per pass: each feeder %.2f s for parsed batches and %.2f s for room in the formatters' queue; the writer %.2f s for the next batch in order\n",
                    g_ns_wait_in.load() * 1e-9 / np / nf, g_ns_wait_out.load() * 1e-9 / np / nf, g_ns_wait_writer.load() * 1e-9 / np);
            fprintf(stderr, "snapgpu-sam: wall per pass if alone: parse %.2f s (%d threads) | feeders (%d): prepare %.2f, align call %.2f, records %.2f, SAM-fields call %.2f | format %.2f s (%d threads) | write %.2f s (1 thread)\n",
                    g_ns_parse.load() * 1e-9 / np / (use_map ? o.n_parse : 1), use_map ? o.n_parse : 1, (int)fctx.size(), g_ns_prep.load() * 1e-9 / np / nf, g_ns_align.load() * 1e-9 / np / nf,
                    g_ns_mid.load() * 1e-9 / np / nf, g_ns_samcall.load() * 1e-9 / np / nf, g_ns_format.load() * 1e-9 / np / o.n_format, o.n_format, g_ns_write.load() * 1e-9 / np);
        }
    }
    return 0;
}
