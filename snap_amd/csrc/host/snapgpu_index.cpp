// snapgpu_index.cpp -- `snapgpu-index <fasta> <output-dir> [options]`: the reference's `snap-aligner index` with the build on the GPU.
//
// Mirrors GenomeIndex::runIndexer (SNAPLib/GenomeIndex.cpp:126-506): same positional arguments, same option spellings for the options that
// make sense here, same four output files -- a directory this writes is what `snap-aligner single <dir> ...`,
// `snap-aligner-gpu` and `snapgpu-sam` load.  All work is behind the C ABI (include/snapgpu.h: snapgpu_index_build_from_fasta,
// snapgpu_built_index_save); this file is option parsing.  Host-only C++; build: g++ -O2 -std=c++17 snapgpu_index.cpp -lsnapgpu
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <chrono>
#include <string>
#include <vector>

#include "../../../include/snapgpu.h"

static void usage()
{
    fprintf(stderr,
            "usage: snapgpu-index <input.fa> <output-dir> [options]\n"
            "  -s <n>              seed length (default 20; the reference's default is 24)\n"
            "  -h <slack>          hash table slack (default 0.3)\n"
            "  -keysize <n>        hash key size in bytes (default: from the seed length, as the reference picks it)\n"
            "  -p<n>               chromosome padding (default 2000)\n"
            "  -B<chars>           contig names end at any of these characters\n"
            "  -bSpace / -bSpace-  contig names end at the first blank or tab (default on)\n"
            "  -AutoAlt-           do not mark *_alt / HLA-* contigs as ALT\n"
            "  -maxAltContigSize <n>, -altContigName <name>, -nonAltContigName <name>, -altContigFile <file>, -nonAltContigFile <file>\n"
            "  -altLiftoverFile <file>\n"
            "  -gpu <n>            HIP device (default 0)\n"
            "  -exact, -t<n>, -q   accepted and ignored (table sizes are always exact; the build is one GPU)\n"
            "  not supported: -large, -locationSize other than 4, -sm, -H<file>  (use the reference's indexer)\n");
    exit(1);
}

static void read_name_file(const char *path, std::vector<std::string> &out)
{
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "unable to open contig list file %s\n", path); exit(1); }
    char *line = nullptr; size_t cap = 0; ssize_t len;
    while ((len = getline(&line, &cap, f)) > 0) {
        while (len > 0 && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
        out.emplace_back(line);
    }
    free(line); fclose(f);
}

int main(int argc, char **argv)
{
    if (argc < 3) usage();
    const char *fasta = argv[1], *out_dir = argv[2];
    snapgpu_index_build_params bp;
    snapgpu_default_index_build_params(&bp);
    std::vector<std::string> alt_names, non_alt_names;
    int device = 0; bool quiet = false;
    for (int n = 3; n < argc; n++) {
        const char *a = argv[n];
        auto need = [&]() -> const char * { if (n + 1 >= argc) usage(); return argv[++n]; };
        if (!strcmp(a, "-s")) bp.seed_len = (uint32_t)atoi(need());
        else if (!strcmp(a, "-h")) bp.slack = atof(need());
        else if (!strcasecmp(a, "-keysize")) bp.key_bytes = (uint32_t)atoi(need());
        else if (!strcasecmp(a, "-locationSize")) { if (atoi(need()) != 4) { fprintf(stderr, "snapgpu-index writes 4-byte genome locations only\n"); return 1; } }
        else if (!strcmp(a, "-exact") || !strcmp(a, "-hg19")) {}
        else if (!strcmp(a, "-q") || !strcmp(a, "-qq")) quiet = true;
        else if (!strcmp(a, "-large") || !strncmp(a, "-sm", 3) || !strncmp(a, "-H", 2)) { fprintf(stderr, "%s is not supported by snapgpu-index (use the reference's indexer)\n", a); return 1; }
        else if (!strcmp(a, "-gpu")) device = atoi(need());
        else if (!strcmp(a, "-bSpace")) bp.space_terminates_name = 1;
        else if (!strcmp(a, "-bSpace-")) bp.space_terminates_name = 0;
        else if (!strcasecmp(a, "-AutoAlt-")) bp.auto_alt = 0;
        else if (!strcmp(a, "-maxAltContigSize")) bp.max_alt_contig_size = atoll(need());
        else if (!strcmp(a, "-altContigName")) alt_names.emplace_back(need());
        else if (!strcmp(a, "-nonAltContigName")) non_alt_names.emplace_back(need());
        else if (!strcmp(a, "-altContigFile")) read_name_file(need(), alt_names);
        else if (!strcmp(a, "-nonAltContigFile")) read_name_file(need(), non_alt_names);
        else if (!strcmp(a, "-altLiftoverFile")) bp.alt_liftover_file = need();
        else if (a[0] == '-' && a[1] == 'B') bp.name_terminators = a + 2;
        else if (a[0] == '-' && a[1] == 'p') { bp.chromosome_padding = (uint32_t)atoi(a + 2); if (!bp.chromosome_padding) { fprintf(stderr, "invalid chromosome padding\n"); return 1; } }
        else if (a[0] == '-' && a[1] == 't') {}
        else if (a[0] == '-' && a[1] == 'O') {}
        else { fprintf(stderr, "Invalid argument: %s\n\n", a); usage(); }
    }
    std::vector<const char *> altp, nonaltp;
    for (auto &s : alt_names) altp.push_back(s.c_str());
    for (auto &s : non_alt_names) nonaltp.push_back(s.c_str());
    bp.alt_contig_names = altp.empty() ? nullptr : altp.data(); bp.n_alt_contig_names = (uint32_t)altp.size();
    bp.non_alt_contig_names = nonaltp.empty() ? nullptr : nonaltp.data(); bp.n_non_alt_contig_names = (uint32_t)nonaltp.size();

    const auto t0 = std::chrono::steady_clock::now();
    snapgpu_built_index *bi = nullptr;
    int rc = snapgpu_index_build_from_fasta(fasta, &bp, device, &bi);
    if (rc != SNAPGPU_OK) { fprintf(stderr, "snapgpu-index: build failed (%d): %s\n", rc, snapgpu_last_error(nullptr)); return 1; }
    const auto t1 = std::chrono::steady_clock::now();
    rc = snapgpu_built_index_save(bi, out_dir);
    if (rc != SNAPGPU_OK) { fprintf(stderr, "snapgpu-index: saving to %s failed (%d): %s\n", out_dir, rc, snapgpu_last_error(nullptr)); return 1; }
    const auto t2 = std::chrono::steady_clock::now();
    snapgpu_index_build_stats st;
    snapgpu_built_index_stats(bi, &st);
    if (!quiet)
        printf("{\"n_bases\": %llu, \"seed_locations\": %llu, \"distinct_seeds\": %llu, \"repeated_seeds\": %llu, \"overflow_table_size\": %llu, "
               "\"hash_table_slots\": %llu, \"hash_blob_bytes\": %llu, \"s_fasta\": %.3f, \"ms_device\": {\"keys\": %.2f, \"sort\": %.2f, \"runs\": %.2f, "
               "\"tables\": %.2f, \"total\": %.2f}, \"s_build_wall\": %.3f, \"s_save\": %.3f}\n",
               (unsigned long long)st.n_bases, (unsigned long long)st.n_seed_locations, (unsigned long long)st.n_distinct_seeds,
               (unsigned long long)st.n_repeated_seeds, (unsigned long long)st.overflow_table_size, (unsigned long long)st.hash_table_slots,
               (unsigned long long)st.hash_blob_bytes, st.s_fasta, st.ms_keys, st.ms_sort, st.ms_runs, st.ms_tables, st.ms_total_device,
               std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count());
    snapgpu_built_index_destroy(bi);
    return 0;
}
