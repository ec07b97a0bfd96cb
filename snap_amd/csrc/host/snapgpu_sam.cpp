// snapgpu_sam.cpp -- FASTQ batcher + SAM writer over the C ABI: the host side of
//     snapgpu-sam single <index-dir> <reads.fq> -o <out.sam> [-d maxDist] [-G-] [-=] [-M] [-Cxx] [-ea] [-D n] [-om n [-omax n] [-mpc n]] [-mrl minReadLength] [-b readsPerBatch]
//     snapgpu-sam paired <index-dir> <reads1.fq> <reads2.fq> -o <out.sam> [same options]
// Streams FASTQ records in batches across include/snapgpu.h -- snapgpu_align_single (BaseAligner::AlignRead) and
// snapgpu_sam_fields_single (what SimpleReadWriter::writeReads / SAMFormat::writeRead compute before they print) -- and prints the
// records the way the reference does.  What is restated here is host-side text handling only:
//   FASTQReader::getReadFromBuffer   SNAPLib/FASTQ.cpp:148-260   (4-line records, '\r' tolerated; plain or gzip input through zlib)
//   Read::clip (ClipBack)            SNAPLib/Read.h:567-620      (the CLI default -C-+: drop the trailing run of '#' qualities)
//   the "useless read" filter        SNAPLib/SingleAligner.cpp:211-232 (dataLength < -mrl or more Ns than -d: written unaligned)
//   SAMFormat::writeHeader           SNAPLib/SAM.cpp:1204-1305   (@HD, default @RG, @PG, one @SQ per contig)
//   SAMFormat::writeRead's snprintf  SNAPLib/SAM.cpp:2078-2098   (field order, PG:Z:SNAP, NM:i, default read-group aux)
//   paired: the both-mates-useless rule (PairedAligner.cpp:680-707), the /1 /2 suffix rule (ReadWriter.cpp:392-404), the mate's quality
//   sum QS:i (SAM.cpp:1826-1837) and writePairs' snprintf (:1855-1877); pairing arithmetic is snapgpu_sam_fields_paired's
// No alignment arithmetic happens on the host: without a GPU snapgpu_create_from_directory fails and so does this program.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <zlib.h>                        // gzopen reads plain and gzip-compressed FASTQ alike (the reference takes .gz input too)

#include "../../../include/snapgpu.h"

static void die(const char *msg, const char *arg = "") { fprintf(stderr, "snapgpu-sam: %s%s\n", msg, arg); exit(1); }

struct Contig { std::string name; uint64_t begin; bool is_alt; };

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
        Contig c; c.begin = (uint64_t)begin; c.is_alt = (cflags & 1) != 0; c.name.assign(line + consumed, (size_t)name_len);
        contigs.push_back(c);
    }
    fclose(f);
}

struct Batch {
    std::vector<std::string> names;
    std::vector<char> bases, quals;          // unclipped, concatenated
    std::vector<uint64_t> offsets;           // n + 1
    void clear() { names.clear(); bases.clear(); quals.clear(); offsets.assign(1, 0); }
};

static bool get_line(gzFile f, std::string &s)
{
    s.clear();
    char buf[4096];
    while (gzgets(f, buf, (int)sizeof(buf)) != NULL) {
        const size_t n = strlen(buf);
        s.append(buf, n);
        if (n > 0 && buf[n - 1] == '\n') { s.pop_back(); if (!s.empty() && s.back() == '\r') s.pop_back(); return true; }
    }
    return !s.empty();
}

// one FASTQ record; false at end of file
static bool next_read(gzFile f, std::string &id, std::string &seq, std::string &qual)
{
    std::string plus;
    if (!get_line(f, id)) return false;
    if (id.empty() || id[0] != '@') die("FASTQ record does not start with '@': ", id.c_str());
    if (!get_line(f, seq) || !get_line(f, plus) || !get_line(f, qual)) die("truncated FASTQ record: ", id.c_str());
    if (plus.empty() || plus[0] != '+') die("FASTQ record without '+' line: ", id.c_str());
    if (seq.size() != qual.size()) die("FASTQ sequence and quality lengths differ: ", id.c_str());
    id.erase(0, 1);
    return true;
}

static char complement(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N'; }      // COMPLEMENT[], Tables.cpp

int main(int argc, char **argv)
{
    const bool paired = argc >= 2 && strcmp(argv[1], "paired") == 0;
    if (argc < (paired ? 5 : 4) || (!paired && strcmp(argv[1], "single") != 0))
        die("usage: snapgpu-sam single <index-dir> <reads.fq> -o <out.sam> | paired <index-dir> <r1.fq> <r2.fq> -o <out.sam>  [-d N] [-G-] [-=] [-M] [-mrl N] [-b N]");
    const std::string index_dir = argv[2], fastq = argv[3], fastq2 = paired ? argv[4] : "";
    std::string out_path;
    snapgpu_params p; snapgpu_default_params(&p);
    p.max_read_len = 400;                                                  // per-wave buffers; SNAPGPU_MAX_READ_LEN raises it (<= 1000), as for snap-aligner-gpu
    if (const char *e = getenv("SNAPGPU_MAX_READ_LEN")) { const int v = atoi(e); if (v >= 50 && v <= 1000) p.max_read_len = (uint32_t)v; }
    bool use_m = true;                                                     // AlignerOptions.cpp:58
    unsigned min_read_len = 50;                                            // -mrl, AlignerOptions.cpp
    size_t batch_reads = 65536;
    bool clip_front = false, clip_back = true;                             // -C-+ (ClipBack) is the default
    int om = -1, mpc = -1; long long omax = 0x7fffffff;
    std::string cl = argv[1];
    for (int i = 2; i < argc; i++) { cl += " "; cl += argv[i]; }
    for (int i = paired ? 5 : 4; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "-o" && i + 1 < argc) out_path = argv[++i];
        else if (a == "-d" && i + 1 < argc) p.max_k = (uint32_t)atoi(argv[++i]);
        else if (a == "-G-") p.use_affine_gap = 0;
        else if (a == "-=") use_m = false;
        else if (a == "-M") use_m = true;
        else if (a.size() == 4 && a.compare(0, 2, "-C") == 0 && strchr("+-", a[2]) && strchr("+-", a[3])) { clip_front = a[2] == '+'; clip_back = a[3] == '+'; }   // AlignerOptions.cpp: -Cxx
        else if (a == "-om" && i + 1 < argc) om = atoi(argv[++i]);                       // secondary alignments (AlignerOptions.cpp:70-72)
        else if (a == "-omax" && i + 1 < argc) omax = atoll(argv[++i]);
        else if (a == "-mpc" && i + 1 < argc) mpc = atoi(argv[++i]);
        else if (a == "-ea") p.emit_alt_alignments = 1;                                  // the first ALT alignment as an extra record (SingleAligner.cpp:320-322)
        else if (a == "-D" && i + 1 < argc) p.extra_search_depth = (uint32_t)atoi(argv[++i]);
        else if (a == "-mrl" && i + 1 < argc) min_read_len = (unsigned)atoi(argv[++i]);
        else if (a == "-b" && i + 1 < argc) batch_reads = (size_t)atoll(argv[++i]);
        else if (a == "-t" && i + 1 < argc) ++i;                           // host threads: nothing to do here
        else die("option not supported: ", a.c_str());
    }
    if (out_path.empty()) die("-o <out.sam> is required");

    std::vector<Contig> contigs; uint64_t n_bases = 0; uint32_t padding = 0;
    load_contigs(index_dir, contigs, n_bases, padding);
    snapgpu_ctx *ctx = NULL;
    int rc = snapgpu_create_from_directory(index_dir.c_str(), &p, 0, &ctx);
    if (rc != SNAPGPU_OK) { fprintf(stderr, "snapgpu-sam: snapgpu_create_from_directory failed (%d): %s\n", rc, snapgpu_last_error(ctx)); return 1; }

    gzFile in = gzopen(fastq.c_str(), "rb");
    if (!in) die("cannot open ", fastq.c_str());
    gzbuffer(in, 1 << 20);
    gzFile in2 = NULL;
    if (paired) {
        in2 = gzopen(fastq2.c_str(), "rb");
        if (!in2) die("cannot open ", fastq2.c_str());
        snapgpu_paired_params pp; snapgpu_default_paired_params(&pp);
        pp.min_read_length = min_read_len;
        rc = snapgpu_enable_paired(ctx, &pp);
        if (rc != SNAPGPU_OK) { fprintf(stderr, "snapgpu-sam: snapgpu_enable_paired failed (%d): %s\n", rc, snapgpu_last_error(ctx)); return 1; }
    }
    if (om >= 0) {
        if (paired) die("-om with `paired` is not supported by this program yet");
        snapgpu_secondary_params sp; memset(&sp, 0, sizeof(sp));
        sp.max_edit_distance = om; sp.max_per_contig = mpc; sp.max_results = omax; sp.adjust_alignments = 0;
        rc = snapgpu_enable_secondary(ctx, &sp);
        if (rc != SNAPGPU_OK) { fprintf(stderr, "snapgpu-sam: snapgpu_enable_secondary failed (%d): %s\n", rc, snapgpu_last_error(ctx)); return 1; }
    }
    FILE *out = fopen(out_path.c_str(), "wb");
    if (!out) die("cannot create ", out_path.c_str());
    // header (SAM.cpp:1232-1295)
    fprintf(out, "@HD\tVN:1.6\tGO:query\n@RG\tID:FASTQ\tPL:Illumina\tPU:pu\tLB:lb\tSM:sm\n@PG\tID:SNAP\tPN:SNAP\tCL:%s\tVN:2.0.5\n", cl.c_str());
    for (size_t c = 0; c < contigs.size(); c++) {
        const uint64_t end = c + 1 < contigs.size() ? contigs[c + 1].begin : n_bases;
        fprintf(out, "@SQ\tSN:%s\tLN:%llu%s\n", contigs[c].name.c_str(), (unsigned long long)(end - contigs[c].begin - padding), contigs[c].is_alt ? "\tAH:*" : "");
    }

    Batch b; b.clear();
    std::string id, seq, qual;
    const uint32_t ops_stride = 64;
    unsigned long long total = 0, aligned = 0;
    bool eof = false;
    while (paired && !eof) {
        b.clear();
        while (b.names.size() < 2 * (batch_reads / 2)) {
            if (!next_read(in, id, seq, qual)) { eof = true; break; }
            std::string id2, seq2, qual2;
            if (!next_read(in2, id2, seq2, qual2)) die("the second FASTQ file has fewer reads than the first");
            if (seq.size() > p.max_read_len || seq2.size() > p.max_read_len) die("read longer than max_read_len (400; set SNAPGPU_MAX_READ_LEN, at most 1000): ", id.c_str());
            b.names.push_back(id); b.bases.insert(b.bases.end(), seq.begin(), seq.end()); b.quals.insert(b.quals.end(), qual.begin(), qual.end()); b.offsets.push_back(b.bases.size());
            b.names.push_back(id2); b.bases.insert(b.bases.end(), seq2.begin(), seq2.end()); b.quals.insert(b.quals.end(), qual2.begin(), qual2.end()); b.offsets.push_back(b.bases.size());
        }
        const size_t n = b.names.size(), np = n / 2;
        if (n == 0) break;
        std::vector<int32_t> front_clip(n, 0), data_len(n, 0);
        std::vector<char> useful(n, 0);
        for (size_t i = 0; i < n; i++) {
            const char *q = b.quals.data() + b.offsets[i], *s = b.bases.data() + b.offsets[i];
            size_t m = (size_t)(b.offsets[i + 1] - b.offsets[i]), fc = 0;
            if (clip_back) while (m > 0 && q[m - 1] == '#') m--;               // Read::clip, Read.h:586-608: back first, then front
            if (clip_front) while (fc < m && q[fc] == '#') fc++;
            front_clip[i] = (int32_t)fc; m -= fc;
            data_len[i] = (int32_t)m;
            unsigned n_count = 0;
            for (size_t j = 0; j < m; j++) n_count += s[fc + j] == 'N';
            useful[i] = m >= min_read_len && n_count <= p.max_k;
        }
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
            for (int w = 0; w < 2; w++) { results[k].status[w] = SNAPGPU_NotFound; results[k].location[w] = SNAPGPU_InvalidGenomeLocation32; results[k].score[w] = -1; }
        }
        if (!to_align.empty()) {
            rc = snapgpu_align_paired(ctx, (uint32_t)to_align.size(), ab.data(), aq.data(), ao.data(), ares.data(), aalt.data());
            if (rc != SNAPGPU_OK) { fprintf(stderr, "snapgpu-sam: snapgpu_align_paired failed (%d): %s\n", rc, snapgpu_last_error(ctx)); return 1; }
            for (size_t k = 0; k < to_align.size(); k++) results[to_align[k]] = ares[k];
        }
        std::vector<int32_t> flag(n), contig(n), mapq(n), n_ops(n), nm(n), stale(n), rnext(n), first_written(np);
        std::vector<int64_t> pos(n), pnext(n), tlen(n);
        std::vector<uint32_t> ops(n * ops_stride);
        rc = snapgpu_sam_fields_paired(ctx, (uint32_t)np, b.bases.data(), b.quals.data(), b.offsets.data(), front_clip.data(), data_len.data(), results.data(),
                                       use_m ? 1 : 0, flag.data(), contig.data(), pos.data(), mapq.data(), ops.data(), ops_stride, n_ops.data(), nm.data(),
                                       rnext.data(), pnext.data(), tlen.data(), first_written.data(), stale.data());
        if (rc != SNAPGPU_OK) { fprintf(stderr, "snapgpu-sam: snapgpu_sam_fields_paired failed (%d): %s\n", rc, snapgpu_last_error(ctx)); return 1; }
        std::string sq, ql;
        for (size_t k = 0; k < np; k++) {
            // QNAME: the /1 /2 suffixes go when both names carry them (ReadWriter.cpp:392-404)
            size_t idl[2] = { b.names[2 * k].size(), b.names[2 * k + 1].size() };
            const std::string &n0 = b.names[2 * k], &n1 = b.names[2 * k + 1];
            if (idl[0] == idl[1] && idl[0] > 2 && n0[idl[0] - 2] == '/' && n1[idl[0] - 2] == '/') {
                const char c0 = n0[idl[0] - 1], c1 = n1[idl[1] - 1];
                if ((c0 == '1' || c0 == '2') && (c1 == '1' || c1 == '2') && c0 != c1) { idl[0] -= 2; idl[1] -= 2; }
            }
            for (int o = 0; o < 2; o++) {
                const int w = o == 0 ? first_written[k] : 1 - first_written[k];
                const size_t i = 2 * k + (size_t)w, im = 2 * k + (size_t)(1 - w);
                const char *s = b.bases.data() + b.offsets[i], *q = b.quals.data() + b.offsets[i];
                const size_t U = (size_t)(b.offsets[i + 1] - b.offsets[i]);
                const std::string &nmq = b.names[i];
                size_t qn = idl[w];
                const size_t sp = nmq.substr(0, qn).find(' ');
                if (sp != std::string::npos) qn = sp;
                sq.assign(s, U); ql.assign(q, U);
                if (flag[i] & 0x10) { for (size_t j = 0; j < U; j++) { sq[U - 1 - j] = complement(s[j]); ql[U - 1 - j] = q[j]; } }
                std::string cigar = "*";
                if (n_ops[i] >= 0) {
                    cigar.clear();
                    char tmp[32];
                    for (int c = 0; c < n_ops[i]; c++) { const uint32_t op = ops[i * ops_stride + (size_t)c]; snprintf(tmp, sizeof(tmp), "%u%c", op >> 4, "MIDNSHP=X"[op & 15]); cigar += tmp; }
                }
                int mqs = 0;                                                // QS: the mate's qualities >= 15, summed (SAM.cpp:1826-1837)
                { const unsigned char *mq = (const unsigned char *)b.quals.data() + b.offsets[im]; const size_t mu = (size_t)(b.offsets[im + 1] - b.offsets[im]);
                  for (size_t j = 0; j < mu; j++) { const int v = (int)mq[j] - '!'; mqs += v >= 15 ? (v != 255) * v : 0; } }
                const char *rn = rnext[i] == -2 ? "=" : (rnext[i] >= 0 ? contigs[(size_t)rnext[i]].name.c_str() : "*");
                fprintf(out, "%.*s\t%d\t%s\t%lld\t%d\t%s\t%s\t%lld\t%d\t%s\t%s\tPG:Z:SNAP\tNM:i:%d\tRG:Z:FASTQ\tPL:Z:Illumina\tPU:Z:pu\tLB:Z:lb\tSM:Z:sm\tQS:i:%d\n",
                        (int)qn, nmq.c_str(), flag[i], contig[i] >= 0 ? contigs[(size_t)contig[i]].name.c_str() : "*", (long long)pos[i], mapq[i], cigar.c_str(),
                        rn, (long long)pnext[i], (int)tlen[i], sq.c_str(), ql.c_str(), nm[i], mqs);
                aligned += (flag[i] & 0x4) == 0;
            }
        }
        total += n;
    }
    while (!paired && !eof) {
        b.clear();
        while (b.names.size() < batch_reads) {
            if (!next_read(in, id, seq, qual)) { eof = true; break; }
            if (seq.size() > p.max_read_len) die("read longer than max_read_len (400; set SNAPGPU_MAX_READ_LEN, at most 1000): ", id.c_str());
            b.names.push_back(id);
            b.bases.insert(b.bases.end(), seq.begin(), seq.end());
            b.quals.insert(b.quals.end(), qual.begin(), qual.end());
            b.offsets.push_back(b.bases.size());
        }
        const size_t n = b.names.size();
        if (n == 0) break;
        // Read::clip (ClipBack) and the useless-read filter; the reads that go to the aligner, clipped, in one buffer
        std::vector<int32_t> front_clip(n, 0), data_len(n, 0);
        std::vector<uint32_t> to_align;
        std::vector<char> ab, aq; std::vector<uint64_t> ao(1, 0);
        for (size_t i = 0; i < n; i++) {
            const char *q = b.quals.data() + b.offsets[i], *s = b.bases.data() + b.offsets[i];
            size_t m = (size_t)(b.offsets[i + 1] - b.offsets[i]), fc = 0;
            if (clip_back) while (m > 0 && q[m - 1] == '#') m--;               // Read::clip, Read.h:586-608: back first, then front
            if (clip_front) while (fc < m && q[fc] == '#') fc++;
            front_clip[i] = (int32_t)fc; m -= fc;
            data_len[i] = (int32_t)m;
            unsigned n_count = 0;
            for (size_t j = 0; j < m; j++) n_count += s[fc + j] == 'N';
            if (m >= min_read_len && n_count <= p.max_k) {
                to_align.push_back((uint32_t)i);
                ab.insert(ab.end(), s + fc, s + fc + m); aq.insert(aq.end(), q + fc, q + fc + m); ao.push_back(ab.size());
            }
        }
        std::vector<snapgpu_single_result> results(n), aligned_res(to_align.size()), alt_res(to_align.size());
        for (size_t i = 0; i < n; i++) {                                   // SingleAligner.cpp:215-225
            memset(&results[i], 0, sizeof(results[i]));
            results[i].status = SNAPGPU_NotFound; results[i].location = SNAPGPU_InvalidGenomeLocation32; results[i].score = -1;
        }
        // records to write: every read's primary, then its secondary results in the aligner's order (SingleAligner.cpp:300-318 -> writeReads)
        std::vector<uint32_t> rec_read;                                     // record -> read of the batch
        std::vector<char> rec_secondary;
        std::vector<snapgpu_single_result> rec_res;
        if (!to_align.empty()) {
            std::vector<snapgpu_single_result> sec; std::vector<uint32_t> nsec(to_align.size(), 0);
            uint32_t stride = 0;
            if (om >= 0) {
                stride = 8;
                for (;;) {
                    sec.assign((size_t)to_align.size() * stride, snapgpu_single_result());
                    rc = snapgpu_align_single_secondary(ctx, (uint32_t)to_align.size(), ab.data(), aq.data(), ao.data(), aligned_res.data(), alt_res.data(),
                                                        sec.data(), stride, nsec.data());
                    if (rc != SNAPGPU_W_SECONDARY_TRUNCATED) break;
                    uint32_t need = stride; for (uint32_t v : nsec) if (v > need) need = v;
                    stride = need;
                }
            } else rc = snapgpu_align_single(ctx, (uint32_t)to_align.size(), ab.data(), aq.data(), ao.data(), aligned_res.data(), alt_res.data());
            if (rc != SNAPGPU_OK) { fprintf(stderr, "snapgpu-sam: alignment failed (%d): %s\n", rc, snapgpu_last_error(ctx)); return 1; }
            std::vector<uint32_t> slot(n, 0xffffffffu);
            for (size_t k = 0; k < to_align.size(); k++) { results[to_align[k]] = aligned_res[k]; slot[to_align[k]] = (uint32_t)k; }
            for (size_t i = 0; i < n; i++) {
                rec_read.push_back((uint32_t)i); rec_secondary.push_back(0); rec_res.push_back(results[i]);
                if (om >= 0 && slot[i] != 0xffffffffu)
                    for (uint32_t j = 0; j < nsec[slot[i]]; j++) { rec_read.push_back((uint32_t)i); rec_secondary.push_back(1); rec_res.push_back(sec[(size_t)slot[i] * stride + j]); }
                if (p.alt_awareness && slot[i] != 0xffffffffu && alt_res[slot[i]].status != SNAPGPU_NotFound) {      // writeReads(&firstALTResult, 1, firstIsPrimary = false)
                    rec_read.push_back((uint32_t)i); rec_secondary.push_back(1); rec_res.push_back(alt_res[slot[i]]);
                }
            }
        } else for (size_t i = 0; i < n; i++) { rec_read.push_back((uint32_t)i); rec_secondary.push_back(0); rec_res.push_back(results[i]); }
        const size_t nr = rec_read.size();
        // the record batch: a read with secondary results appears once per record
        std::vector<char> rb, rq; std::vector<uint64_t> ro(1, 0); std::vector<int32_t> rfc(nr), rdl(nr);
        for (size_t r = 0; r < nr; r++) {
            const size_t i = rec_read[r];
            rb.insert(rb.end(), b.bases.begin() + (long)b.offsets[i], b.bases.begin() + (long)b.offsets[i + 1]);
            rq.insert(rq.end(), b.quals.begin() + (long)b.offsets[i], b.quals.begin() + (long)b.offsets[i + 1]);
            ro.push_back(rb.size()); rfc[r] = front_clip[i]; rdl[r] = data_len[i];
        }
        std::vector<int32_t> flag(nr), contig(nr), mapq(nr), n_ops(nr), nm(nr), stale(nr);
        std::vector<int64_t> pos(nr);
        std::vector<uint32_t> ops(nr * ops_stride);
        rc = snapgpu_sam_fields_single(ctx, (uint32_t)nr, rb.data(), rq.data(), ro.data(), rfc.data(), rdl.data(), rec_res.data(), use_m ? 1 : 0,
                                       flag.data(), contig.data(), pos.data(), mapq.data(), ops.data(), ops_stride, n_ops.data(), nm.data(), stale.data());
        if (rc != SNAPGPU_OK) { fprintf(stderr, "snapgpu-sam: snapgpu_sam_fields_single failed (%d): %s\n", rc, snapgpu_last_error(ctx)); return 1; }
        for (size_t r = 0; r < nr; r++) if (rec_secondary[r]) flag[r] |= 0x100;                  // SAM_SECONDARY (createSAMLine, SAM.cpp:1477-1479)
        std::string sq, ql;
        for (size_t i = 0; i < nr; i++) {
            const size_t rd = rec_read[i];
            const char *s = b.bases.data() + b.offsets[rd], *q = b.quals.data() + b.offsets[rd];
            const size_t U = (size_t)(b.offsets[rd + 1] - b.offsets[rd]);
            const std::string &nmq = b.names[rd];
            const size_t sp = nmq.find(' ');                               // "illegal in SAM: truncate at the space" (SAM.cpp:2001-2004)
            sq.assign(s, U); ql.assign(q, U);
            if (flag[i] & 0x10) { for (size_t j = 0; j < U; j++) { sq[U - 1 - j] = complement(s[j]); ql[U - 1 - j] = q[j]; } }
            std::string cigar = "*";
            if (n_ops[i] >= 0) {
                cigar.clear();
                char tmp[32];
                for (int k = 0; k < n_ops[i]; k++) {
                    const uint32_t op = ops[i * ops_stride + (size_t)k];
                    snprintf(tmp, sizeof(tmp), "%u%c", op >> 4, "MIDNSHP=X"[op & 15]);
                    cigar += tmp;
                }
            }
            fprintf(out, "%.*s\t%d\t%s\t%lld\t%d\t%s\t*\t0\t0\t%s\t%s\tPG:Z:SNAP\tNM:i:%d\tRG:Z:FASTQ\tPL:Z:Illumina\tPU:Z:pu\tLB:Z:lb\tSM:Z:sm\n",
                    (int)(sp == std::string::npos ? nmq.size() : sp), nmq.c_str(), flag[i], contig[i] >= 0 ? contigs[(size_t)contig[i]].name.c_str() : "*",
                    (long long)pos[i], mapq[i], cigar.c_str(), sq.c_str(), ql.c_str(), nm[i]);
            aligned += (flag[i] & 0x4) == 0;
        }
        total += n;
    }
    fclose(out); gzclose(in); if (in2) gzclose(in2);
    snapgpu_destroy(ctx);
    fprintf(stderr, "snapgpu-sam: %llu reads, %llu mapped records\n", total, aligned);
    return 0;
}
