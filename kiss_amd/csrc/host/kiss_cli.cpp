// kiss_cli.cpp -- `kiss`: the kISS command line on top of libkiss_hip.so (host C++17, no HIP, no Boost).
//
// Keeps the reference's CLI surface (reference include/utils/options.hpp:83-203, src/main.cpp:19-39):
//   kiss [-h] [-v] [-g] [-t NUM] [--verbose] <command> [options] <FASTA filename/Text filename>
//   suffix_sort    [-k NUM(=256, -1 = unbounded)] [-s PARALLEL_SORTING|PREFIX_DOUBLING]   (command/suffix_sort.hpp)
//   fmindex_build  [-k NUM (ignored, like the reference)]  -> writes <fasta>.fmi            (command/fmindex_build.hpp)
//   fmindex_query  [-q STR] [-n NUM(=10)] [-b patterns.bin]                                  (command/fmindex_query.hpp)
// and its log fields ("n = …, k = …, suffix sorting elapsed …", "query = … found N times", "searching time",
// "number of matched locations", "location checksum").  Extras (opt-in): --output-sa FILE (raw u32 LE, n+1
// entries; the reference never writes the SA), --device N, and for suffix_sort --gpus N / --devices LIST (the LMS sort
// sharded over several GPUs of the node by ONE process, kiss_hip_multi_*; include/kiss_hip.h).
// Input handling as utils/io.hpp:6-18 + suffix_sort.hpp:33: FASTA if the first byte is '>' (all records
// concatenated), else plain text lines; ACGT/acgt -> 0..3, every other character -> 4 % 4 = 0 (A).
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/kiss_hip.h"

namespace {

const char *VERSION = "1.0.0-hip";

void usage()
{
    std::cerr << "kiss " << VERSION << " (MI355X / libkiss_hip)\n"
              << "kiss [--generic-option ...] cmd [--cmd-specific-option ...]\n"
              << "Generic options:\n"
              << "  -h [ --help ]            produce help message\n"
              << "  -v [ --version ]         print version string\n"
              << "  -g [ --generic ]         (not supported) generic alphabet\n"
              << "  -t [ --num_threads ] NUM accepted for compatibility; no result depends on it\n"
              << "  --verbose                print per-stage device times\n"
              << "  --device NUM             HIP device index (default 0)\n\n"
              << "./kiss suffix_sort [--option ...] <FASTA filename/Text filename>\n"
              << "  -k [ --kordered ] NUM (=256)   k-ordered value; -1 indicates unbounded sorting\n"
              << "  -s [ --sorting-algorithm ] ALGO (=PARALLEL_SORTING)   PARALLEL_SORTING or PREFIX_DOUBLING\n"
              << "  --output-sa FILE               also write the suffix array (raw uint32 LE, n+1 entries)\n"
              << "  --gpus NUM (=1)                shard the LMS sort over NUM devices (--device, --device + 1, ...)\n"
              << "  --devices LIST                 the same with an explicit comma-separated device list\n\n"
              << "./kiss fmindex_build [--option ...] <FASTA filename/Text filename>\n"
              << "  -k [ --kordered ] NUM (=256)   accepted and ignored (the index is built with k = 32)\n\n"
              << "./kiss fmindex_query [--option ...] <FASTA filename/Text filename>\n"
              << "  -q [ --query ] STR             content of the query string\n"
              << "  -n [ --headn ] NUM (=10)       output the first n locations in single query mode\n"
              << "  -b [ --batch ] patterns.bin    batch query mode (u32 len, u32 count, then count x len bytes)\n";
}

inline uint8_t to_code(unsigned char c)
{
    switch (c) {
    case 'a': case 'A': return 0;
    case 'c': case 'C': return 1;
    case 'g': case 'G': return 2;
    case 't': case 'T': return 3;
    default: return 0; // Codec::to_int gives 4, the commands apply % 4 (suffix_sort.hpp:33)
    }
}

void check(int rc, const char *where)
{
    if (rc != KISS_HIP_OK) throw std::runtime_error(std::string(where) + ": " + kiss_hip_strerror(rc));
}

// utils/io.hpp:6-18 (read_sequence) + the `% 4` of the commands: the file is streamed to the device and parsed there
// (kiss_amd/csrc/fasta.hip); the host never touches a base
struct DeviceText {
    kiss_hip_ctx *ctx = nullptr;
    uint8_t *d_S = nullptr;
    uint64_t n = 0;
    double create_s = 0, load_s = 0;
    bool own_ctx = true;
    // borrowed != nullptr: load through a context that something else owns (the first device's context of a
    // kiss_hip_multi: one set of work arrays serves the loader and the sort; allocating a second set after freeing the
    // first costs seconds on this driver -- profiles/r03_first_call_allocation_times_*.log)
    DeviceText(const std::string &path, int device, kiss_hip_ctx *borrowed = nullptr)
    {
        uint64_t bytes = 0;
        if (kiss_hip_file_size(path.c_str(), &bytes) != KISS_HIP_OK) throw std::runtime_error("cannot open " + path);
        auto t0 = std::chrono::steady_clock::now();
        if (borrowed) {
            ctx = borrowed;
            own_ctx = false;
        } else {
            check(kiss_hip_ctx_create(&ctx, device, bytes ? bytes : 1), "kiss_hip_ctx_create");
        }
        create_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        t0 = std::chrono::steady_clock::now();
        const int rc = kiss_hip_ctx_load_text_file(ctx, path.c_str(), &d_S, &n);
        load_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rc != KISS_HIP_OK) {
            if (own_ctx) kiss_hip_ctx_destroy(ctx);
            throw std::runtime_error(std::string("kiss_hip_ctx_load_text_file: ") + kiss_hip_strerror(rc));
        }
    }
    std::vector<uint8_t> to_host() const
    {
        std::vector<uint8_t> S(n);
        check(kiss_hip_copy_to_host(S.data(), d_S, n), "kiss_hip_copy_to_host");
        return S;
    }
    ~DeviceText()
    {
        kiss_hip_free_dev(d_S);
        if (ctx && own_ctx) kiss_hip_ctx_destroy(ctx);
    }
    DeviceText(const DeviceText &) = delete;
    DeviceText &operator=(const DeviceText &) = delete;
};

struct Args {
    std::string command, fasta, query, batch, output_sa, algo = "PARALLEL_SORTING";
    long long k = 256;
    size_t headn = 10;
    int device = 0, gpus = 1;
    std::vector<int> devices; // --devices; empty: device, device + 1, ... (gpus of them)
    bool verbose = false, generic = false;
};

Args parse(int argc, char **argv)
{
    Args a;
    std::vector<std::string> pos;
    for (int i = 1; i < argc; i++) {
        std::string s = argv[i];
        auto next = [&](const char *name) -> std::string {
            if (i + 1 >= argc) throw std::runtime_error(std::string("the required argument for option '") + name + "' is missing");
            return argv[++i];
        };
        if (s == "-h" || s == "--help") { usage(); std::exit(1); }
        else if (s == "-v" || s == "--version") { std::cerr << VERSION << std::endl; std::exit(1); }
        else if (s == "-g" || s == "--generic") a.generic = true;
        else if (s == "--verbose") a.verbose = true;
        else if (s == "-t" || s == "--num_threads") (void)next("--num_threads");
        else if (s == "--device") a.device = std::stoi(next("--device"));
        else if (s == "-k" || s == "--kordered") a.k = std::stoll(next("--kordered"));
        else if (s == "-s" || s == "--sorting-algorithm") a.algo = next("--sorting-algorithm");
        else if (s == "-q" || s == "--query") a.query = next("--query");
        else if (s == "-n" || s == "--headn") a.headn = (size_t)std::stoull(next("--headn"));
        else if (s == "-b" || s == "--batch") a.batch = next("--batch");
        else if (s == "--output-sa") a.output_sa = next("--output-sa");
        else if (s == "--gpus") a.gpus = std::stoi(next("--gpus"));
        else if (s == "--devices") {
            const std::string list = next("--devices");
            size_t at = 0;
            while (at <= list.size()) {
                const size_t comma = list.find(',', at);
                const std::string item = list.substr(at, comma == std::string::npos ? std::string::npos : comma - at);
                if (item.empty()) throw std::runtime_error("--devices: empty entry in '" + list + "'");
                a.devices.push_back(std::stoi(item));
                if (comma == std::string::npos) break;
                at = comma + 1;
            }
        }
        else if (!s.empty() && s[0] == '-' && s.size() > 1) throw std::runtime_error("unrecognised option '" + s + "'");
        else pos.push_back(s);
    }
    if (pos.empty()) { usage(); std::exit(1); }
    if (a.gpus < 1) throw std::runtime_error("--gpus must be >= 1");
    if (a.devices.empty())
        for (int g = 0; g < a.gpus; g++) a.devices.push_back(a.device + g);
    else
        a.device = a.devices[0]; // the text is loaded, and the induction runs, on the first device of the list
    a.command = pos[0];
    if (pos.size() < 2) throw std::runtime_error("the option '--fasta' is required but missing");
    a.fasta = pos[1];
    std::transform(a.algo.begin(), a.algo.end(), a.algo.begin(), [](unsigned char c) { return (char)std::toupper(c); });
    return a;
}

double seconds_since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// ---- .fmi (fm_index.hpp:591-646; Serializer: u64 count + raw bytes, nothing when the count is 0) ----------
struct Fmi {
    kiss_hip_fmi_sizes z{};
    uint32_t cnt[4]{}, pri = 0;
    std::vector<uint8_t> bwt, occ2;
    std::vector<uint32_t> occ1, sa, b_occ;
    std::vector<uint64_t> b;
    void alloc(uint64_t n)
    {
        check(kiss_hip_fmi_sizes_for(n, &z), "kiss_hip_fmi_sizes_for");
        bwt.assign(z.bwt_bytes, 0);
        occ1.assign(z.occ1_entries, 0);
        occ2.assign(z.occ2_bytes, 0);
        sa.assign(z.sa_entries, 0);
        b.assign(z.b_words, 0);
        b_occ.assign(z.b_occ_entries, 0);
    }
    static void put(std::ofstream &o, uint64_t count, const void *p, uint64_t bytes)
    {
        if (!count) return;
        o.write(reinterpret_cast<const char *>(&count), 8);
        o.write(reinterpret_cast<const char *>(p), (std::streamsize)bytes);
    }
    void save(const std::string &path) const
    {
        std::ofstream o(path, std::ios::binary);
        if (!o) throw std::runtime_error("cannot write " + path);
        o.write(reinterpret_cast<const char *>(cnt), 16);
        o.write(reinterpret_cast<const char *>(&pri), 4);
        put(o, z.n_sa, bwt.data(), z.bwt_bytes);
        put(o, z.occ1_entries / 4, occ1.data(), z.occ1_entries * 4);
        put(o, z.occ2_bytes / 4, occ2.data(), z.occ2_bytes);
        put(o, z.sa_entries, sa.data(), z.sa_entries * 4);
        const uint32_t lookup[2] = {0u, (uint32_t)z.n_sa}; // LOOKUP_LEN = 0 (fm_index.hpp:238-258)
        put(o, 2, lookup, 8);
        put(o, z.n_sa, b.data(), z.b_words * 8);
        put(o, z.b_occ_entries, b_occ.data(), z.b_occ_entries * 4);
    }
    static uint64_t get_count(std::ifstream &in)
    {
        uint64_t c = 0;
        in.read(reinterpret_cast<char *>(&c), 8);
        if (!in) throw std::runtime_error("truncated .fmi");
        return c;
    }
    void load(const std::string &path)
    {
        std::ifstream in(path, std::ios::binary);
        if (!in) throw std::runtime_error("cannot open " + path + " (run fmindex_build first)");
        in.read(reinterpret_cast<char *>(cnt), 16);
        in.read(reinterpret_cast<char *>(&pri), 4);
        const uint64_t N = get_count(in);
        alloc(N - 1);
        in.read(reinterpret_cast<char *>(bwt.data()), (std::streamsize)z.bwt_bytes);
        if (get_count(in) * 4 != z.occ1_entries) throw std::runtime_error("bad occ1 size in .fmi");
        in.read(reinterpret_cast<char *>(occ1.data()), (std::streamsize)(z.occ1_entries * 4));
        if (get_count(in) * 4 != z.occ2_bytes) throw std::runtime_error("bad occ2 size in .fmi");
        in.read(reinterpret_cast<char *>(occ2.data()), (std::streamsize)z.occ2_bytes);
        if (get_count(in) != z.sa_entries) throw std::runtime_error("bad sa size in .fmi");
        in.read(reinterpret_cast<char *>(sa.data()), (std::streamsize)(z.sa_entries * 4));
        if (get_count(in) != 2) throw std::runtime_error("bad lookup size in .fmi (LOOKUP_LEN = 0 expected)");
        uint32_t lookup[2];
        in.read(reinterpret_cast<char *>(lookup), 8);
        if (get_count(in) != N) throw std::runtime_error("bad b_ size in .fmi");
        in.read(reinterpret_cast<char *>(b.data()), (std::streamsize)(z.b_words * 8));
        if (get_count(in) != z.b_occ_entries) throw std::runtime_error("bad b_occ_ size in .fmi");
        in.read(reinterpret_cast<char *>(b_occ.data()), (std::streamsize)(z.b_occ_entries * 4));
        if (!in || in.peek() != EOF) throw std::runtime_error("trailing or missing bytes in .fmi"); // fm_index.hpp:642
    }
    kiss_hip_fmi_view view() const
    {
        kiss_hip_fmi_view v{};
        v.n_sa = z.n_sa;
        for (int c = 0; c < 4; c++) v.cnt[c] = cnt[c];
        v.pri = pri;
        v.sa_intv = 4;
        v.bwt = bwt.data();
        v.occ1 = occ1.data();
        v.occ2 = occ2.data();
        v.sa = sa.data();
        v.b = b.data();
        v.b_occ = b_occ.data();
        return v;
    }
};

int suffix_sort_main(const Args &a)
{
    const bool multi = a.devices.size() > 1;
    kiss_hip_multi *mc = nullptr;
    double multi_create_s = 0;
    if (multi) { // one process, several devices: the per-device contexts come first, the first one also loads the file
        // Two distinct GPUs exchange the whole LMS list and gather half of the sorted list over the ONE xGMI link between
        // them: the phase model (bench.py: scaling_model) puts a chm13-size sort at about 97 ms on two GPUs
        // against 80 ms on one.  The result is the same; the user is told that the second GPU does not pay.
        if (a.devices.size() == 2 && a.devices[0] != a.devices[1])
            std::fprintf(stderr, "[warning] --gpus 2: two GPUs share one xGMI link for the exchange of the LMS list and the gather of the "
                                 "sorted pieces; the phase model expects this to be SLOWER than one GPU (about 0.8x; 4 GPUs: about "
                                 "1.4x, 8 GPUs: about 2.1x).  Same result either way.\n");
        uint64_t bytes = 0;
        if (kiss_hip_file_size(a.fasta.c_str(), &bytes) != KISS_HIP_OK) throw std::runtime_error("cannot open " + a.fasta);
        const auto tc = std::chrono::steady_clock::now();
        check(kiss_hip_multi_create(&mc, a.devices.data(), (int)a.devices.size(), bytes ? bytes : 1), "kiss_hip_multi_create");
        multi_create_s = seconds_since(tc);
    }
    struct McGuard {
        kiss_hip_multi *&m;
        ~McGuard()
        {
            if (m) kiss_hip_multi_destroy(m);
        }
    } mc_guard{mc};
    DeviceText T(a.fasta, a.device, multi ? kiss_hip_multi_ctx(mc, 0) : nullptr);
    int algo;
    if (a.algo == "PARALLEL_SORTING") algo = KISS_HIP_ALGO_PARALLEL_SORTING;
    else if (a.algo == "PREFIX_DOUBLING") algo = KISS_HIP_ALGO_PREFIX_DOUBLING;
    else throw std::invalid_argument("Invalid sorting algorithm");
    const uint32_t k = (uint32_t)(uint64_t)a.k; // -1 -> size_t max -> truncated to 0xFFFFFFFF (suffix_sort.hpp:35-37)
    void *d_SA = nullptr;
    const auto ta = std::chrono::steady_clock::now();
    check(kiss_hip_alloc_dev(&d_SA, (T.n + 1) * sizeof(uint32_t)), "kiss_hip_alloc_dev");
    const double alloc_s = seconds_since(ta);
    const auto t0 = std::chrono::steady_clock::now(); // the reference starts its stopwatch here (suffix_sort.hpp:57)
    if (multi)
        check(kiss_hip_multi_suffix_sort_dna_u32_dev(mc, T.d_S, T.n, k, algo, (uint32_t *)d_SA),
              "kiss_hip_multi_suffix_sort_dna_u32_dev");
    else
        check(kiss_hip_ctx_suffix_sort_dna_u32_dev(T.ctx, T.d_S, T.n, k, algo, (uint32_t *)d_SA, nullptr),
              "kiss_hip_ctx_suffix_sort_dna_u32_dev");
    const double el = seconds_since(t0);
    std::fprintf(stderr, "[info] n = %llu, k = %llu, suffix sorting elapsed %.6f\n", (unsigned long long)T.n,
                 (unsigned long long)(a.k < 0 ? ~0ull : (unsigned long long)a.k), el);
    if (multi && a.verbose) {
        kiss_hip_multi_stats ms;
        kiss_hip_multi_get_stats(mc, &ms);
        std::fprintf(stderr, "[debug] %u devices (workspaces %.6f s): pack + broadcast %.3f ms, get_lms per slice %.3f ms, "
                             "partition %.3f ms, exchange %.3f ms, lms_suffix_direct_sort per key range %.3f ms, gather %.3f ms, "
                             "put_lms_suffix + induced_sort %.3f ms, total %.3f ms; lms = %llu\n",
                     ms.ndev, multi_create_s, ms.ms_pack, ms.ms_classify, ms.ms_partition, ms.ms_exchange, ms.ms_sort,
                     ms.ms_gather, ms.ms_induce, ms.ms_total, (unsigned long long)ms.m);
    }
    if (!multi && a.verbose) {
        kiss_hip_stats st;
        kiss_hip_get_stats(T.ctx, &st);
        std::fprintf(stderr, "[debug] device workspace %.6f s; read + upload + device-side parse of %s %.6f s; SA buffer %.6f s\n",
                     T.create_s, a.fasta.c_str(), T.load_s, alloc_s);
        std::fprintf(stderr,
                     "[debug] device: pack %.3f ms, get_lms %.3f ms, lms_suffix_direct_sort %.3f ms, put_lms_suffix %.3f ms, "
                     "induced_sort %.3f ms, prefix_doubling %.3f ms, total %.3f ms; lms = %llu, rounds = %u + %u, passes = %u\n",
                     st.ms_pack, st.ms_classify, st.ms_lms_sort, st.ms_place, st.ms_induce, st.ms_refine, st.ms_total,
                     (unsigned long long)st.m, st.lms_rounds, st.doubling_rounds, st.induce_passes);
    }
    if (!a.output_sa.empty()) {
        std::ofstream o(a.output_sa, std::ios::binary);
        if (!o) throw std::runtime_error("cannot write " + a.output_sa);
        const uint64_t total = T.n + 1, chunk = 64ull << 20; // entries per piece
        std::vector<uint32_t> buf((size_t)std::min<uint64_t>(total, chunk));
        for (uint64_t off = 0; off < total; off += chunk) {
            const uint64_t c = std::min<uint64_t>(chunk, total - off);
            check(kiss_hip_copy_to_host(buf.data(), (const uint32_t *)d_SA + off, c * sizeof(uint32_t)), "kiss_hip_copy_to_host");
            o.write(reinterpret_cast<const char *>(buf.data()), (std::streamsize)(c * sizeof(uint32_t)));
        }
    }
    const auto tf = std::chrono::steady_clock::now();
    kiss_hip_free_dev(d_SA);
    if (a.verbose) std::fprintf(stderr, "[debug] SA buffer released in %.6f s\n", seconds_since(tf));
    return 0;
}

int fmindex_build_main(const Args &a)
{
    std::vector<uint8_t> S;
    {
        DeviceText T(a.fasta, a.device);
        S = T.to_host();
    }
    if (S.empty()) throw std::runtime_error("empty sequence");
    Fmi f;
    f.alloc(S.size());
    check(kiss_hip_fmi_build_host(S.data(), S.size(), nullptr, f.bwt.data(), f.occ1.data(), f.occ2.data(), f.sa.data(),
                                  f.b.data(), f.b_occ.data(), f.cnt, &f.pri, a.device),
          "kiss_hip_fmi_build_host");
    f.save(a.fasta + ".fmi");
    return 0;
}

const char *ending(size_t x)
{
    x %= 100;
    if (x / 10 == 1) return "th";
    if (x % 10 == 1) return "st";
    if (x % 10 == 2) return "nd";
    if (x % 10 == 3) return "rd";
    return "th";
}

int fmindex_query_main(const Args &a)
{
    std::vector<uint8_t> S;
    {
        DeviceText T(a.fasta, a.device);
        S = T.to_host();
    }
    Fmi f;
    f.load(a.fasta + ".fmi");
    const kiss_hip_fmi_view v = f.view();
    if (!a.query.empty()) {
        std::vector<uint8_t> q;
        for (unsigned char c : a.query) q.push_back(to_code(c));
        uint32_t beg = 0, end = 0;
        uint64_t hits = 0, chk = 0;
        check(kiss_hip_fmi_query_batch_host(&v, q.data(), (uint32_t)q.size(), 1, &beg, &end, &hits, &chk, nullptr, nullptr, 0,
                                            a.device), "kiss_hip_fmi_query_batch_host");
        std::vector<uint32_t> off(hits + 1);
        std::vector<uint64_t> idx(2);
        if (hits)
            check(kiss_hip_fmi_query_batch_host(&v, q.data(), (uint32_t)q.size(), 1, &beg, &end, &hits, &chk, off.data(),
                                                idx.data(), hits, a.device), "kiss_hip_fmi_query_batch_host");
        std::string qs;
        for (auto c : q) qs.push_back("ACGT"[c]);
        std::fprintf(stderr, "[info] query = %s found %llu times\n", qs.c_str(), (unsigned long long)hits);
        for (size_t i = 0; i < std::min<size_t>(a.headn, hits); i++) {
            std::string sub;
            for (size_t j = 0; j < q.size() && off[i] + j < S.size(); j++) sub.push_back("ACGT"[S[off[i] + j]]);
            std::fprintf(stderr, "[info] The %zu-%s position is %u, content of substring is %s\n", i + 1, ending(i + 1), off[i],
                         sub.c_str());
        }
    }
    if (!a.batch.empty()) {
        std::ifstream p(a.batch, std::ios::binary);
        if (!p) throw std::runtime_error("cannot open " + a.batch);
        uint32_t L = 0, Q = 0;
        p.read(reinterpret_cast<char *>(&L), 4);
        p.read(reinterpret_cast<char *>(&Q), 4);
        std::fprintf(stderr, "[info] query_len: %u, num_query: %u\n", L, Q);
        std::vector<uint8_t> pat((size_t)L * Q);
        p.read(reinterpret_cast<char *>(pat.data()), (std::streamsize)pat.size());
        if (!p) throw std::runtime_error("truncated pattern file");
        for (auto &c : pat) c = to_code(c);
        std::vector<uint32_t> beg(Q), end(Q);
        uint64_t hits = 0, chk = 0;
        const auto t0 = std::chrono::steady_clock::now();
        check(kiss_hip_fmi_query_batch_host(&v, pat.data(), L, Q, beg.data(), end.data(), &hits, &chk, nullptr, nullptr, 0,
                                            a.device), "kiss_hip_fmi_query_batch_host");
        std::fprintf(stderr, "[info] searching time: %.6f seconds\n", seconds_since(t0));
        std::fprintf(stderr, "[info] number of matched locations: %llu\n", (unsigned long long)hits);
        std::fprintf(stderr, "[info] location checksum: %llu\n", (unsigned long long)chk);
    }
    return 0;
}

} // namespace

int main(int argc, char **argv)
{
    try {
        Args a = parse(argc, argv);
        // TODO of the reference kept as is: generic alphabets are rejected (suffix_sort.hpp:26-28)
        if (a.generic) throw std::invalid_argument("Generic sorting and indexing are currently not supported.");
        if (a.command == "suffix_sort") return suffix_sort_main(a);
        if (a.command == "fmindex_build") return fmindex_build_main(a);
        if (a.command == "fmindex_query") return fmindex_query_main(a);
        usage();
        throw std::runtime_error("invalid command '" + a.command + "'");
    } catch (const std::exception &e) {
        std::cerr << e.what() << std::endl;
        return 1;
    }
}
