// demux.cpp -- `fqtk demux`: host pipeline around the MI355X barcode matcher.
//
// the callers and data formats either side of the hot path.  Mirrors
// Demux::execute (/root/reference/src/bin/commands/demux.rs:881-1001) with the same flags
// (:597-652), validation messages (:806-875), file naming (:674-688), skip rule (:300-306,954-957),
// record routing (:968-975), header rewriting (:171-267) and metrics file (:994-998).
//
// What is different, by design (MI355X-first; nothing here is translated from the Rust):
//   * the matcher is NOT called per template: templates are gathered into chunks, their sample
//     barcodes packed into a pinned SoA buffer (demux.rs:121-123 defines the concatenation) and
//     matched on the GPU through the C ABI (include/fqtk_match.h) on alternating pipeline slots, so
//     gunzip/parse of chunk k+1, H2D+kernel+D2H of chunk k and BGZF-compress/write of chunk k-1 overlap;
//   * routing/compression is partitioned BY SAMPLE across worker threads (each output file has one
//     owner, no locks, input order preserved per file as in the sequential reference loop).
// All matching goes through libfqtk_match.so; there is no CPU matching path here.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <malloc.h>
#include <sys/prctl.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>
#include <csignal>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <string_view>
#include <thread>
#include <vector>

#include "../../../include/fqtk_demux.h"
#include "../../../include/fqtk_match.h"
#include "bgzf.hpp"
#include "bgzf_walk.hpp"
#include "chunk_dispatch.hpp"
#include "chunk_schedule.hpp"
#include "fastq_io.hpp"
#include "header.hpp"
#include "metrics.hpp"
#include "read_structure.hpp"
#include "region_inflate.hpp"
#include "samples.hpp"

using namespace fqtk_host;

namespace {

double now_s() {
    static const auto t0 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

void info(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    std::fprintf(stderr, "[%8.3f INFO fqtk] ", now_s());
    std::vfprintf(stderr, fmt, ap);
    std::fputc('\n', stderr);
    va_end(ap);
}

// Output files this run has created.  A fatal error (from any thread) must not leave truncated BGZF files
// behind -- a .fq.gz without its EOF block looks complete to most tools -- so die() removes them before
// the process ends.  (The reference panics; whatever its pooled writers had flushed stays on disk.)
std::mutex g_created_mu;
std::vector<std::string> g_created;
std::atomic<bool> g_dying{false};

[[noreturn]] void die(const std::string &msg) {
    if (g_dying.exchange(true))          // another thread is already reporting: let it finish, never return
        for (;;) pause();
    std::fprintf(stderr, "Error: %s\n", msg.c_str());
    {
        std::lock_guard<std::mutex> lk(g_created_mu);
        for (const std::string &p : g_created) unlink(p.c_str());   // open handles of other threads stay valid
        if (!g_created.empty())
            std::fprintf(stderr, "Error: removed %zu partially written output file(s)\n", g_created.size());
    }
    std::fflush(stderr);
    std::_Exit(1);
}

// Everything is on disk and closed.  The process ends here: tearing the HIP runtime down through destructors and
// atexit handlers costs ~0.25 s and frees nothing the OS does not free.  FQTK_CLEAN_EXIT=1 takes the long way (tools
// that write their output at exit, like rocprofv3, need it).
// ... and even `_Exit` takes the kernel 0.1-0.5 s: the process's pages behind the HIP runtime's queues (2.2 GB), its page-locked buffers (0.2 ms
// per MB to let go, as to lock) and 20 GB of device mappings are taken apart before `wait()` returns to whoever started the run -- a fifth of a
// 64 M-template run's wall clock (round 5: "after the last line 0.25-0.5 s").  So the run happens in a CHILD of the process the user started
// (main() forks before anything else exists); when every file is closed the child says so down a pipe, closes its standard streams and ends, and the
// parent returns at once with the child's status while the kernel clears up behind it.  A run that fails, or is killed, ends the old way: the parent
// waits for the child and passes its status on.  FQTK_FOREGROUND=1 (and FQTK_CLEAN_EXIT=1, which tools that write at exit need): no child.
int g_done_fd = -1;
[[noreturn]] void end_process() {
    std::fflush(stdout);
    std::fflush(stderr);
    if (env_on("FQTK_CLEAN_EXIT")) std::exit(0);
    if (g_done_fd >= 0) {
        const unsigned char ok = 0;
        if (::write(g_done_fd, &ok, 1) == 1) { ::close(g_done_fd); ::close(0); ::close(1); ::close(2); }
    }
    std::_Exit(0);
}

pid_t g_child = -1;
void forward_signal(int sig) { if (g_child > 0) ::kill(g_child, sig); }
// The parent's whole life: wait for the child's "done" byte (-> 0) or for its end (-> its status).
[[noreturn]] void supervise(pid_t child, int fd) {
    g_child = child;
    for (int sig : {SIGINT, SIGTERM, SIGHUP, SIGQUIT}) {
        struct sigaction sa;
        std::memset(&sa, 0, sizeof sa);
        sa.sa_handler = forward_signal;
        sigaction(sig, &sa, nullptr);
    }
    unsigned char b = 0;
    ssize_t r;
    do r = ::read(fd, &b, 1); while (r < 0 && errno == EINTR);
    if (r == 1) std::_Exit((int)b);
    int st = 0;
    pid_t w;
    do w = ::waitpid(child, &st, 0); while (w < 0 && errno == EINTR);
    if (w == child && WIFEXITED(st)) std::_Exit(WEXITSTATUS(st));
    if (w == child && WIFSIGNALED(st)) { signal(WTERMSIG(st), SIG_DFL); ::raise(WTERMSIG(st)); std::_Exit(128 + WTERMSIG(st)); }
    std::_Exit(1);
}

bool g_timing = false;   // FQTK_TIMING: clocks of the stages in the log
// (FQTK_TIMING: where a run's first second goes -- page-locking costs 0.2-0.4 s per GB, and while one call is at it every other HIP call of the process waits.
//  Measured at the end of round 5: a first buffer of the first stretch's size and the full-size ones made by the prefetching thread cut the first chunk by 40 ms and
//  cost the steady rate 4 % -- the same wall clock; not kept.)
static int pinned_alloc_timed(size_t bytes, void **out) {
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = fqtk_pinned_alloc(bytes, out);
    if (g_timing && bytes >= (8u << 20))
        std::fprintf(stderr, "(timing) page-locked allocation of %zu MB: %.1f ms, done at epoch %.3f\n", bytes >> 20, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(),
                     std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count());
    return rc;
}

// (FQTK_TIMING: the anonymous resident memory right now, in MB -- pinned staging and what the HIP runtime keeps on the host)
size_t rss_anon_mb() {
    size_t kb = 0;
    if (FILE *f = std::fopen("/proc/self/status", "r")) {
        char buf[256];
        while (std::fgets(buf, sizeof buf, f))
            if (!std::strncmp(buf, "RssAnon:", 8)) kb = (size_t)std::strtoull(buf + 8, nullptr, 10);
        std::fclose(f);
    }
    return kb >> 10;
}

// (end of a run: what it cost the host)
void report_footprint(size_t n_files) {
    rusage ru;
    if (getrusage(RUSAGE_SELF, &ru) == 0)
        info("Host footprint: peak resident set %.1f MB, %zu output files open at once.", ru.ru_maxrss / 1024.0, n_files);
    if (g_timing) {   // what the kernel has to take apart when the process ends, and when that starts (wall clock)
        std::string line;
        if (FILE *f = std::fopen("/proc/self/status", "r")) {
            char buf[256];
            while (std::fgets(buf, sizeof buf, f))
                if (!std::strncmp(buf, "VmRSS", 5) || !std::strncmp(buf, "RssAnon", 7) || !std::strncmp(buf, "RssFile", 7) || !std::strncmp(buf, "RssShmem", 8) || !std::strncmp(buf, "VmLck", 5) || !std::strncmp(buf, "VmPin", 5)) {
                    std::string t(buf);
                    while (!t.empty() && (t.back() == '\n' || t.back() == ' ')) t.pop_back();
                    for (char &c : t) if (c == '\t') c = ' ';
                    line += (line.empty() ? "" : "; ") + t;
                }
            std::fclose(f);
        }
        timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);
        info("(timing) at the end: %s; epoch %.3f", line.c_str(), ts.tv_sec + ts.tv_nsec * 1e-9);
    }
}

struct Options {
    std::vector<std::string> inputs, read_structures, skip_reasons;
    std::vector<char> output_types{'T'};
    std::string sample_metadata, output, unmatched_prefix = "unmatched";
    unsigned long max_mismatches = 1, min_mismatch_delta = 2, threads = 8, compression_level = 5;
    int device = 0;
    std::vector<int> devices;        // --devices a,b,..: chunk k goes to devices[k mod G]
    unsigned long chunk_reads = 1ul << 17;
    bool chunk_given = false;
    bool host_output = false;        // --host-output: format and compress the records on the host (the reference's way)
    bool host_inflate = false;       // --host-inflate: BGZF inputs are inflated by the reader threads even where the device could
    bool gpu_gunzip = false;         // --gpu-gunzip: single-stream gzip inputs are decoded on the device in chunks
};

const char *kUsage =
    "Usage: fqtk demux [OPTIONS] --inputs <INPUTS>... --read-structures <READ_STRUCTURES>... \\\n"
    "                  --sample-metadata <SAMPLE_METADATA> --output <OUTPUT>\n"
    "  -i, --inputs <INPUTS>...                    input FASTQ files, one per sequencing read\n"
    "  -r, --read-structures <READ_STRUCTURES>...  read structures, one per input FASTQ\n"
    "  -b, --output-types <OUTPUT_TYPES>...        segment types to write: T B M C [default: T]\n"
    "  -s, --sample-metadata <SAMPLE_METADATA>     TSV with columns sample_id, barcode\n"
    "  -o, --output <OUTPUT>                       output directory\n"
    "  -u, --unmatched-prefix <PREFIX>             [default: unmatched]\n"
    "      --max-mismatches <N>                    [default: 1]\n"
    "  -d, --min-mismatch-delta <N>                [default: 2]\n"
    "  -t, --threads <N>                           [default: 8] (must be 5 or more)\n"
    "  -c, --compression-level <N>                 [default: 5]\n"
    "  -S, --skip-reasons <REASON>...              too-few-bases\n"
    "      --device <N>                            GPU to use [default: 0] (additive flag)\n"
    "      --devices <A,B,..>                      several GPUs: chunk k is matched, formatted and compressed on devices[k mod G] (additive flag).\n"
    "                                              Compressed inputs stay on the devices: every input is inflated on one of them (largest file first, to the\n"
    "                                              least loaded), and a chunk's text goes from there to the chunk's device over xGMI\n"
    "      --chunk-reads <N>                       templates per GPU chunk [default: 262144; 131072 with --host-output] (additive flag)\n"
    "      --host-output                           parse, format and BGZF-compress the records on the host CPUs (as the\n"
    "                                              reference does) instead of on the GPU, which is the default: there the\n"
    "                                              inputs' text goes to the device, records are formatted and DEFLATE-compressed\n"
    "                                              in HBM and whole BGZF members come back (additive flag; alias --no-gpu-bgzf;\n"
    "                                              --gpu-bgzf is accepted and means the default)\n"
    "      --host-inflate                          inflate compressed inputs on the host CPUs.  Without it, when every input is compressed:\n"
    "                                              BGZF members (bgzip, htslib, fqtk's own outputs) go to the device as they\n"
    "                                              are, a wavefront each; serial gzip files (gzip, bcl2fastq), from 64 MB of them, in chunks of\n"
    "                                              64 KiB cut at DEFLATE block starts the device finds, windows handed down the chain; what cannot\n"
    "                                              be cut or decoded like that is decoded by a host thread -- any valid file is read (additive flag)\n"
    "      --gpu-gunzip                            serial gzip inputs on the device whatever their size (additive flag)\n";

bool parse_ulong(const std::string &s, unsigned long *out) {
    if (s.empty()) return false;
    char *end = nullptr;
    unsigned long v = std::strtoul(s.c_str(), &end, 10);
    if (*end != '\0' || s[0] == '-') return false;
    *out = v;
    return true;
}

Options parse_args(int argc, char **argv) {
    Options o;
    std::vector<std::string> args(argv, argv + argc);
    size_t i = 0;
    auto canon = [](const std::string &a) -> std::string {
        static const std::pair<const char *, const char *> shorts[] = {
            {"-i", "--inputs"}, {"-r", "--read-structures"}, {"-b", "--output-types"}, {"-s", "--sample-metadata"},
            {"-o", "--output"}, {"-u", "--unmatched-prefix"}, {"-d", "--min-mismatch-delta"}, {"-t", "--threads"},
            {"-c", "--compression-level"}, {"-S", "--skip-reasons"}};
        for (auto &p : shorts) if (a == p.first) return p.second;
        return a;
    };
    bool types_given = false;
    while (i < args.size()) {
        std::string a = args[i++];
        std::string inline_val;
        bool has_inline = false;
        const size_t eq = a.find('=');
        if (a.rfind("--", 0) == 0 && eq != std::string::npos) {
            inline_val = a.substr(eq + 1);
            a = a.substr(0, eq);
            has_inline = true;
        }
        a = canon(a);
        auto multi = [&](std::vector<std::string> &dst) {
            if (has_inline) { dst.push_back(inline_val); return; }
            size_t start = dst.size();
            while (i < args.size() && !(args[i].size() > 1 && args[i][0] == '-' && !std::isdigit((unsigned char)args[i][1]) && args[i] != "-"))
                dst.push_back(args[i++]);
            if (dst.size() == start) die("a value is required for '" + a + "' but none was supplied");
        };
        auto single = [&]() -> std::string {
            if (has_inline) return inline_val;
            if (i >= args.size()) die("a value is required for '" + a + "' but none was supplied");
            return args[i++];
        };
        auto num = [&](unsigned long *dst) {
            const std::string v = single();
            if (!parse_ulong(v, dst)) die("invalid value '" + v + "' for '" + a + "': invalid digit found in string");
        };
        if (a == "--inputs") multi(o.inputs);
        else if (a == "--read-structures") multi(o.read_structures);
        else if (a == "--output-types") {
            std::vector<std::string> v;
            multi(v);
            if (!types_given) o.output_types.clear();
            types_given = true;
            for (const std::string &t : v) {
                if (t.size() != 1) die("invalid value '" + t + "' for '--output-types': too many characters in string");
                o.output_types.push_back(t[0]);
            }
        } else if (a == "--sample-metadata") o.sample_metadata = single();
        else if (a == "--output") o.output = single();
        else if (a == "--unmatched-prefix") o.unmatched_prefix = single();
        else if (a == "--max-mismatches") num(&o.max_mismatches);
        else if (a == "--min-mismatch-delta") num(&o.min_mismatch_delta);
        else if (a == "--threads") num(&o.threads);
        else if (a == "--compression-level") num(&o.compression_level);
        else if (a == "--skip-reasons") multi(o.skip_reasons);
        else if (a == "--device") { unsigned long d; num(&d); o.device = (int)d; }
        else if (a == "--devices") {
            const std::string v = single();
            size_t p = 0;
            while (p <= v.size()) {
                const size_t q = v.find(',', p);
                const std::string tok = v.substr(p, q == std::string::npos ? std::string::npos : q - p);
                unsigned long d;
                if (!parse_ulong(tok, &d)) die("invalid value '" + v + "' for '--devices': expected a comma-separated list of device indices");
                o.devices.push_back((int)d);
                if (q == std::string::npos) break;
                p = q + 1;
            }
        }
        else if (a == "--chunk-reads") { num(&o.chunk_reads); o.chunk_given = true; }
        else if (a == "--gpu-bgzf") o.host_output = false;
        else if (a == "--host-output" || a == "--no-gpu-bgzf") o.host_output = true;
        else if (a == "--host-inflate") o.host_inflate = true;
        else if (a == "--gpu-gunzip") o.gpu_gunzip = true;
        else if (a == "--help" || a == "-h") { std::fputs(kUsage, stdout); std::exit(0); }
        else die("unexpected argument '" + a + "' found\n\n" + kUsage);
    }
    std::string missing;
    if (o.inputs.empty()) missing += " --inputs <INPUTS>...";
    if (o.read_structures.empty()) missing += " --read-structures <READ_STRUCTURES>...";
    if (o.sample_metadata.empty()) missing += " --sample-metadata <SAMPLE_METADATA>";
    if (o.output.empty()) missing += " --output <OUTPUT>";
    if (!missing.empty()) die("the following required arguments were not provided:" + missing + "\n\n" + kUsage);
    return o;
}

template <typename T>
class BoundedQueue {
  public:
    explicit BoundedQueue(size_t cap) : cap_(cap) {}
    void push(T v) {
        std::unique_lock<std::mutex> lk(mu_);
        if (q_.size() >= cap_) ++full_waits;
        not_full_.wait(lk, [&] { return q_.size() < cap_; });
        q_.push_back(std::move(v));
        not_empty_.notify_one();
    }
    bool try_push(T &v) {   // false (v untouched) when the queue is full
        std::unique_lock<std::mutex> lk(mu_);
        if (q_.size() >= cap_) return false;
        q_.push_back(std::move(v));
        not_empty_.notify_one();
        return true;
    }
    T pop() {
        std::unique_lock<std::mutex> lk(mu_);
        not_empty_.wait(lk, [&] { return !q_.empty(); });
        T v = std::move(q_.front());
        q_.pop_front();
        not_full_.notify_one();
        return v;
    }
    uint64_t full_waits = 0;   // pushes that found the queue full (FQTK_TIMING report)
  private:
    size_t cap_;
    std::mutex mu_;
    std::condition_variable not_full_, not_empty_;
    std::deque<T> q_;
};

struct ReadResult {   // one batch from one input, or an error
    std::unique_ptr<RecBatch> batch;
    std::string error;
};

struct Chunk {
    std::vector<std::unique_ptr<RecBatch>> batches;   // one per input
    size_t n = 0;
    std::vector<uint8_t> skip;          // per template
    std::vector<fqtk_match_t> res;      // per template (skipped ones hold NO_MATCH and are never routed)
};

struct SegRef { uint32_t input, seg; };

// One BGZF output file, owned by exactly one router thread.
struct OutFile {
    FILE *f = nullptr;
    std::string path;
    std::string buf;                 // router side: uncompressed tail, < one block
    uint64_t next_submit = 0;        // router side: sequence number of the next block handed out
    std::mutex mu;                   // writer side: blocks may finish out of order
    uint64_t next_write = 0;
    std::map<uint64_t, std::vector<uint8_t>> ready;
};

struct StageTimes {   // FQTK_TIMING=1: where the host threads spend their time (seconds, summed over threads)
    std::atomic<uint64_t> router_wait{0}, router_format{0}, router_submit{0}, comp_wait{0}, comp_deflate{0}, comp_write{0};
    std::atomic<uint64_t> submit_calls{0}, submit_cut{0}, submit_push{0}, submit_slab{0};
    std::atomic<uint64_t> main_wait{0}, main_gpu_wait{0}, main_handoff{0}, reader_parse{0}, reader_push{0};
};
StageTimes g_times;
inline uint64_t tick() { return g_timing ? (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0; }

struct CompressJob {
    OutFile *of = nullptr;           // nullptr = shut down
    uint64_t seq = 0;
    std::string data;
};

struct Plan {
    std::vector<ReadStructure> rs;
    std::vector<SegRef> by_type[4];   // T, B, M, C in (input, segment) order
    bool want[4] = {false, false, false, false};
    size_t files_per_sample = 0;
    size_t file_base[4] = {0, 0, 0, 0};
};

const SegType kTypes[4] = {SegType::Template, SegType::SampleBarcode, SegType::MolecularBarcode, SegType::CellularBarcode};
const char kCodes[4] = {'R', 'I', 'U', 'C'};

// Block buffers travel router -> compressor -> back: allocating and freeing a 64 KiB string per block
// across threads was the routers' largest cost (glibc arena traffic), so they are recycled.
class BufferPool {
  public:
    std::string get() {
        std::lock_guard<std::mutex> lk(mu_);
        if (free_.empty()) { std::string s; s.reserve(kBgzfBlockSize + 4096); return s; }
        std::string s = std::move(free_.back());
        free_.pop_back();
        return s;
    }
    void put(std::string &&s) {
        s.clear();
        std::lock_guard<std::mutex> lk(mu_);
        if (free_.size() < 4096) free_.push_back(std::move(s));
    }
  private:
    std::mutex mu_;
    std::vector<std::string> free_;
};
BufferPool g_pool;

// Router side: cut full 65280-byte blocks (or the final partial one) off the file's buffer and hand
// them to the compression pool.  Sequence numbers keep the file's block order.  The file's buffer itself
// becomes the job (no copy of the block): only the few hundred bytes past the cut move to a fresh buffer.
using JobQueues = std::vector<std::unique_ptr<BoundedQueue<CompressJob>>>;   // one per compressor thread
void submit_blocks(OutFile &of, JobQueues &jobs, bool final) {
    while (of.buf.size() >= kBgzfBlockSize || (final && !of.buf.empty())) {
        const uint64_t ta = tick();
        const size_t n = std::min(kBgzfBlockSize, of.buf.size());
        std::string rest = g_pool.get();
        rest.assign(of.buf, n, std::string::npos);
        of.buf.resize(n);
        CompressJob j;
        j.of = &of;
        j.seq = of.next_submit++;
        j.data = std::move(of.buf);
        of.buf = std::move(rest);
        // one queue per compressor (a single shared queue was a lock convoy at 30+ threads); blocks of a
        // file go round-robin over them, so uneven sample sizes do not unbalance the compressors
        const size_t q = (reinterpret_cast<uintptr_t>(&of) / sizeof(OutFile) + j.seq) % jobs.size();
        const uint64_t tb = tick();
        // work-conserving: a compressor stalled in a write must not hold up the router while others idle
        bool placed = false;
        for (size_t k = 0; k < jobs.size() && !placed; ++k) placed = jobs[(q + k) % jobs.size()]->try_push(j);
        if (!placed) jobs[q]->push(std::move(j));
        if (g_timing) { g_times.submit_calls += 1; g_times.submit_cut += tb - ta; g_times.submit_push += tick() - tb; }
    }
}

// Pool side: compress one block, then write it -- and any successors already waiting -- in order.
void compress_and_write(CompressJob &j, BlockCompressor &bc) {
    std::vector<uint8_t> comp;
    std::string err;
    const uint64_t t0 = tick();
    if (!bc.compress(reinterpret_cast<const uint8_t *>(j.data.data()), j.data.size(), comp, &err)) die(err);
    const uint64_t t1 = tick();
    g_times.comp_deflate += t1 - t0;
    OutFile &of = *j.of;
    std::lock_guard<std::mutex> lk(of.mu);
    of.ready.emplace(j.seq, std::move(comp));
    for (auto it = of.ready.begin(); it != of.ready.end() && it->first == of.next_write; it = of.ready.erase(it)) {
        if (std::fwrite(it->second.data(), 1, it->second.size(), of.f) != it->second.size()) die("write failed: " + of.path);
        ++of.next_write;
    }
    g_times.comp_write += tick() - t1;
}


// ---- the GPU record pipeline (include/fqtk_demux.h): the default output path ------------------------------------------
// The host keeps what is I/O: a reader thread per input copies the text of the next `chunk` records into page-locked
// memory (decompressing if need be; a record is four lines, so it only counts newlines), this thread hands the chunks
// to the device -- where the records are indexed, matched, formatted into per-file 65 280-byte blocks and DEFLATE-
// compressed -- and a collector appends the BGZF members that come back to the output files.
struct PinnedRaw : RawBuffer {
    ~PinnedRaw() override { if (data) fqtk_pinned_free(data); }
    bool grow(size_t want, size_t keep) override {
        void *p = nullptr;
        if (pinned_alloc_timed(want, &p) != FQTK_OK) return false;
        if (keep) std::memcpy(p, data, keep);
        if (data) fqtk_pinned_free(data);
        data = static_cast<char *>(p);
        cap = want;
        return true;
    }
};
struct RawChunk { PinnedRaw *buf = nullptr; size_t n = 0, bytes = 0; std::string error; };

// A second pair of hands for one memcpy: a large plain input is copied into page-locked memory by two threads at once
// (one thread moves ~10 GB/s out of the page cache; the two 150-base files of a run need twice that to keep the device fed).
class CopyHelper {
  public:
    CopyHelper() : th_([this] { run(); }) {}
    ~CopyHelper() { { std::lock_guard<std::mutex> lk(mu_); stop_ = true; } cv_.notify_all(); th_.join(); }
    void start(char *dst, const char *src, size_t n) {
        { std::lock_guard<std::mutex> lk(mu_); dst_ = dst; src_ = src; n_ = n; busy_ = true; }
        cv_.notify_all();
    }
    void wait() { std::unique_lock<std::mutex> lk(mu_); cv_.wait(lk, [&] { return !busy_; }); }
  private:
    void run() {
        for (;;) {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return stop_ || busy_; });
            if (stop_) return;
            lk.unlock();
            std::memcpy(dst_, src_, n_);
            lk.lock();
            busy_ = false;
            cv_.notify_all();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    char *dst_ = nullptr;
    const char *src_ = nullptr;
    size_t n_ = 0;
    bool busy_ = false, stop_ = false;
    std::thread th_;
};

void write_all(int fd, const uint8_t *p, size_t n, const std::string &path) {
    while (n) {
        const ssize_t w = ::write(fd, p, n);
        if (w < 0) { if (errno == EINTR) continue; die("write failed: " + path + ": " + std::strerror(errno)); }
        p += w;
        n -= (size_t)w;
    }
}

// Which configurations the device pipeline takes (the limits of include/fqtk_demux.h); anything else is formatted on the host.
bool gpu_output_supported(const Plan &plan, std::string *why) {
    if (plan.rs.size() > FQTK_DEMUX_MAX_INPUTS) { *why = "more than " + std::to_string(FQTK_DEMUX_MAX_INPUTS) + " inputs"; return false; }
    if (plan.files_per_sample > FQTK_DEMUX_MAX_FILES) { *why = "more than " + std::to_string(FQTK_DEMUX_MAX_FILES) + " output files per sample"; return false; }
    if (plan.by_type[1].size() + plan.by_type[2].size() > 23) { *why = "more than 23 sample / molecular barcode segments"; return false; }
    return true;
}

[[noreturn]] void run_gpu_output(const Options &opt, const Plan &plan, const std::vector<Sample> &samples,
                                 std::vector<std::unique_ptr<FastqSource>> &sources, bool skip_few) {
    const size_t n_inputs = plan.rs.size(), S = samples.size(), G = opt.devices.size();
    size_t chunk = std::max<unsigned long>(1, opt.chunk_given ? opt.chunk_reads : 262144ul);
    std::vector<size_t> per_record_of(n_inputs, 0);   // bytes of text per record, judged by every input's first MiB
    {   // a chunk's text must stay below 2 GiB per input (32-bit offsets on the device): long reads get smaller chunks
        size_t per_record = 0;
        for (size_t i = 0; i < sources.size(); ++i) {
            per_record_of[i] = sources[i]->estimate_raw_bytes(1024) / 1024;
            per_record = std::max(per_record, per_record_of[i]);
        }
        const size_t fit = per_record ? std::max<size_t>(1, (768ull << 20) / per_record) : chunk;
        if (fit < chunk) {
            info("Records of about %zu bytes: %zu templates per chunk instead of %zu.", per_record, fit, chunk);
            chunk = fit;
        }
    }
    const uint32_t L = (uint32_t)samples[0].barcode.size();

    // ---- BGZF inputs: their members go to a device compressed and are inflated there (fqtk_demuxer_feed), when every input is
    // one.  With several devices every input has a HOME device -- the one that inflates it and keeps its text --, the chunks
    // are cut out of the homes' texts in order and chunk k's windows are copied to device k mod G over xGMI where they are
    // not at home there (include/fqtk_demux.h: fqtk_demuxer_fed_cut / fqtk_demuxer_submit_windows)
    std::vector<std::unique_ptr<BgzfFile>> bgzf_in;   // (the mapped file of every fed input, BGZF or serial gzip)
    std::vector<char> is_serial_gz(n_inputs, 0);
    // serial gzip inputs go to the device when asked for (--gpu-gunzip), or by themselves when there is enough of them for the chunks to
    // fill it (64 MB of .gz in all; below that the host's decoders are done before the device's buffers are allocated)
    bool gpu_gunzip = opt.gpu_gunzip || env_on("FQTK_GPU_GUNZIP");
    if (!gpu_gunzip && !env_on("FQTK_NO_GPU_GUNZIP")) {
        uint64_t gz_bytes = 0;
        for (size_t i = 0; i < n_inputs; ++i) {
            struct stat st;
            if (sources[i]->kind() == FastqSource::Kind::Gzip && stat(opt.inputs[i].c_str(), &st) == 0 && S_ISREG(st.st_mode)) gz_bytes += (uint64_t)st.st_size;
        }
        gpu_gunzip = gz_bytes >= (64ull << 20);
        // ... and only files the chunks can be cut out of: a dynamic-Huffman block must start somewhere in the second and third MiB
        // of every one of them (a file of stored blocks, or of blocks of many MB, stays with the host's decoders, which need no cuts)
        for (size_t i = 0; i < n_inputs && gpu_gunzip; ++i) {
            if (sources[i]->kind() != FastqSource::Kind::Gzip) continue;
            BgzfFile probe;
            std::string e;
            if (!probe.open(opt.inputs[i], &e)) { gpu_gunzip = false; break; }
            if (probe.size > (4u << 20)) {
                SpecInflate finder;
                finder.attach(probe.map, probe.size);
                if (finder.find_block_start(8ull << 20, 24ull << 20) == ~0ull) {
                    info("No DEFLATE block starts in the second and third MiB of %s: gzip inputs are decoded on the host.", opt.inputs[i].c_str());
                    gpu_gunzip = false;
                }
            }
        }
    }
    bool fed_mode = !opt.host_inflate && !env_on("FQTK_HOST_INFLATE");
    size_t n_serial = 0;
    for (size_t i = 0; i < n_inputs && fed_mode; ++i) {
        const FastqSource::Kind kd = sources[i]->kind();
        if (kd == FastqSource::Kind::Gzip && gpu_gunzip) { is_serial_gz[i] = 1; ++n_serial; }
        else if (kd != FastqSource::Kind::Bgzf) { fed_mode = false; break; }
        std::string e;
        bgzf_in.push_back(std::make_unique<BgzfFile>());
        if (!bgzf_in.back()->open(opt.inputs[i], &e)) fed_mode = false;   // (a pipe: the reader threads inflate it)
    }
    // the inputs' homes: largest file first, each to the device with the fewest bytes so far (one device: all on it)
    std::vector<size_t> home_of(n_inputs, 0);
    if (!fed_mode) {
        bgzf_in.clear();
        n_serial = 0;
    } else {
        std::vector<size_t> order(n_inputs);
        for (size_t i = 0; i < n_inputs; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return bgzf_in[a]->size > bgzf_in[b]->size; });
        std::vector<uint64_t> load(G, 0);
        for (size_t i : order) {
            size_t best = 0;
            for (size_t g = 1; g < G; ++g) if (load[g] < load[best]) best = g;
            home_of[i] = best;
            load[best] += (uint64_t)bgzf_in[i]->size + 1;
        }
        if (n_serial) info("gzip inputs: decoded on the device in chunks (BGZF members: one wavefront each).");
        else info("BGZF inputs: members are inflated on the device.");
        if (G > 1) {
            std::string where;
            for (size_t i = 0; i < n_inputs; ++i) where += (i ? ", " : "") + std::to_string(opt.devices[home_of[i]]);
            info("%zu devices: input i is inflated on device [%s] and its text stays there; chunk k's text goes to device k mod %zu from device to device.", G, where.c_str(), G);
        }
    }

    // ---- devices: matcher + record pipeline each (their bring-up overlaps the first reads and the file creation)
    std::vector<fqtk_matcher *> matchers(G, nullptr);
    std::vector<fqtk_demuxer *> demuxers(G, nullptr);
    std::vector<const char *> bc, ids;
    for (const Sample &s : samples) { bc.push_back(s.barcode.c_str()); ids.push_back(s.sample_id.c_str()); }
    std::vector<uint32_t> n_segments;
    std::vector<fqtk_demux_segment> segments;
    for (const ReadStructure &r : plan.rs) {
        n_segments.push_back((uint32_t)r.segments.size());
        for (const ReadSegment &g : r.segments)
            segments.push_back(fqtk_demux_segment{(uint32_t)g.offset, g.has_length() ? (int32_t)g.length : -1, (char)g.kind});
    }
    // (several devices: each is brought up by a thread of its own -- a device's context, code objects, memo and pipeline buffers take 0.17-0.22 s, and
    //  eight of them one after another were 1.5 s before the first chunk; the first device alone first: it initialises the runtime)
    auto bring_up = [&](size_t g) {
            if (fqtk_matcher_create(bc.data(), (uint32_t)S, L, (uint8_t)opt.max_mismatches, (uint8_t)opt.min_mismatch_delta, opt.devices[g], &matchers[g]) != FQTK_OK)
                die(std::string("cannot create the GPU barcode matcher: ") + fqtk_last_error());
            fqtk_matcher_set_sample_ids(matchers[g], ids.data());
            if (g_timing) info("(timing) matcher on device %d created; anonymous resident memory %zu MB.", opt.devices[g], rss_anon_mb());
            fqtk_demux_config cfg;
            std::memset(&cfg, 0, sizeof cfg);
            cfg.n_inputs = (uint32_t)n_inputs;
            cfg.n_segments = n_segments.data();
            cfg.segments = segments.data();
            for (int k = 0; k < 4; ++k) cfg.want[k] = plan.want[k] ? 1 : 0;
            cfg.skip_too_few_bases = skip_few ? 1 : 0;
            cfg.max_chunk_templates = (uint32_t)std::min<size_t>(chunk, 1u << 22);
            cfg.carry_blocks = G == 1 ? 1 : 0;   // several devices take the chunks in turn: a file's stream must not wait on any of them
            cfg.compression_level = (int)opt.compression_level;
            if (fqtk_demuxer_create(matchers[g], &cfg, &demuxers[g]) != FQTK_OK) die(std::string("cannot set up the GPU record pipeline: ") + fqtk_last_error());
            info("GPU barcode matcher and record pipeline ready on device %d (%llu memo entries).", opt.devices[g],
                 (unsigned long long)fqtk_matcher_memo_entries(matchers[g]));
            if (g_timing) info("(timing) anonymous resident memory: %zu MB.", rss_anon_mb());
    };
    std::thread gpu_init([&] {
        bring_up(0);
        std::vector<std::thread> rest;
        for (size_t g = 1; g < G; ++g) rest.emplace_back(bring_up, g);
        for (auto &t : rest) t.join();
    });

    // ---- readers
    // page-locked staging per input: one buffer being filled, one waiting, and one per device whose H2D copy may still be
    // reading it (fqtk_demuxer_text_done)
    const size_t kRing = 2 + G;
    std::vector<std::unique_ptr<BoundedQueue<RawChunk>>> rq;
    std::vector<std::unique_ptr<BoundedQueue<PinnedRaw *>>> free_bufs;
    std::vector<std::vector<std::unique_ptr<PinnedRaw>>> rings(n_inputs);
    for (size_t i = 0; i < n_inputs; ++i) {
        rq.push_back(std::make_unique<BoundedQueue<RawChunk>>(kRing));
        free_bufs.push_back(std::make_unique<BoundedQueue<PinnedRaw *>>(kRing));
        for (size_t k = 0; k < kRing; ++k) {
            rings[i].push_back(std::make_unique<PinnedRaw>());
            // two to begin with: page-locking memory goes through the HIP runtime, where the device bring-up is busy
            // (a dozen 90 MB allocations queued up in front of the matcher's own calls); the third joins when it is done
            if (k < 2) free_bufs[i]->push(rings[i].back().get());
        }
    }
    std::vector<std::thread> readers;
    // A large plain (mapped) input: one thread only COUNTS newlines to cut the chunks, a second one copies every cut into
    // page-locked memory together with a helper, while the next cut is being counted.  Everything else (small files,
    // gzip / BGZF / pipes): one thread that decodes, counts and copies (FastqSource::next_raw).
    std::vector<std::unique_ptr<BoundedQueue<std::pair<FastqSource::RawCut, std::string>>>> cuts(n_inputs);
    std::vector<std::vector<std::unique_ptr<CopyHelper>>> helpers(n_inputs);
    const bool split_ok = usable_cpus() >= 12 && !env_on("FQTK_NO_SPLIT_READERS");
    // helpers per large input: one with --threads 16 (the fixed threads -- readers, writers, this one, the collector -- take 13
    // of them on a four-file run), more when --threads and the machine leave room
    size_t n_large = 0;
    for (size_t i = 0; i < n_inputs; ++i) n_large += split_ok && sources[i]->mapped() && sources[i]->mapped_size() >= (1ull << 30);
    size_t n_helpers = 1;
    if (n_large) {
        const size_t have = std::min<size_t>(opt.threads, usable_cpus()), fixed = n_inputs + 2 * n_large + 4 + 3;
        n_helpers = std::min<size_t>(3, std::max<size_t>(1, have > fixed ? (have - fixed) / n_large : 1));
    }
    // ... and a second thread that counts newlines for the cutter (of the fixed threads the writers, the collector and the
    // unmapper mostly wait)
    const bool count_assist = n_large && std::min<size_t>(opt.threads, usable_cpus()) >= 16 && !env_on("FQTK_NO_COUNT_ASSISTANT");
    if (const char *e = std::getenv("FQTK_COPY_HELPERS")) n_helpers = std::min<size_t>(8, std::max<size_t>(1, (size_t)std::atoi(e)));
    for (size_t i = 0; i < n_inputs && !fed_mode; ++i) {
        if (split_ok && sources[i]->mapped() && sources[i]->mapped_size() >= (1ull << 30)) {
            cuts[i] = std::make_unique<BoundedQueue<std::pair<FastqSource::RawCut, std::string>>>(2);
            for (size_t h = 0; h < n_helpers; ++h) helpers[i].push_back(std::make_unique<CopyHelper>());
            if (count_assist) sources[i]->attach_count_assistant();
            readers.emplace_back([&, i] {   // the cutter
                for (;;) {
                    std::pair<FastqSource::RawCut, std::string> c;
                    const uint64_t t0 = tick();
                    const bool ok = sources[i]->next_cut(std::min<size_t>(chunk, 1u << 22), &c.first, &c.second);
                    g_times.reader_parse += tick() - t0;
                    if (!ok && c.second.empty()) c.second = "read failed";
                    const bool last = !ok || c.first.n_records == 0;
                    cuts[i]->push(std::move(c));
                    if (last) return;
                }
            });
            readers.emplace_back([&, i] {   // the copier
                for (;;) {
                    auto c = cuts[i]->pop();
                    RawChunk out;
                    out.error = c.second;
                    out.n = c.first.n_records;
                    out.bytes = c.first.bytes + (c.first.add_newline ? 1 : 0);
                    const bool last = !out.error.empty() || out.n == 0;
                    if (!last) {
                        out.buf = free_bufs[i]->pop();
                        const uint64_t t0 = tick();
                        if (out.buf->cap < out.bytes + 1 && !out.buf->grow(out.bytes + out.bytes / 16 + 65536, 0)) die(std::string("cannot allocate page-locked memory: ") + fqtk_last_error());
                        const size_t parts = helpers[i].size() + 1, part = (c.first.bytes / parts) & ~(size_t)63;
                        for (size_t h = 0; h < helpers[i].size(); ++h) {
                            const size_t from = part * (h + 1), upto = h + 2 == parts ? c.first.bytes : part * (h + 2);
                            helpers[i][h]->start(out.buf->data + from, c.first.p + from, upto - from);
                        }
                        std::memcpy(out.buf->data, c.first.p, part);
                        for (auto &h : helpers[i]) h->wait();
                        sources[i]->release_cut(c.first);   // only now may the cutter unmap it (fastq_io.hpp)
                        if (c.first.add_newline) out.buf->data[c.first.bytes] = '\n';
                        g_times.reader_push += tick() - t0;   // (the copy: reported as "push")
                    } else {
                        out.buf = nullptr;
                    }
                    rq[i]->push(std::move(out));
                    if (last) return;
                }
            });
            continue;
        }
        readers.emplace_back([&, i] {
            for (;;) {
                RawChunk c;
                c.buf = free_bufs[i]->pop();
                const uint64_t t0 = tick();
                if (c.buf->cap == 0) {   // first use: sized by what the input's first lines look like, so that it need not grow
                    const size_t want = sources[i]->estimate_raw_bytes(std::min<size_t>(chunk, 1u << 22));
                    if (want && !c.buf->grow(want, 0)) die(std::string("cannot allocate page-locked memory: ") + fqtk_last_error());
                    if (g_timing) info("(timing) input %zu: %zu MB of page-locked memory ready.", i, want >> 20);
                }
                const bool ok = sources[i]->next_raw(std::min<size_t>(chunk, 1u << 22), c.buf, &c.n, &c.bytes, &c.error);
                g_times.reader_parse += tick() - t0;
                if (!ok && c.error.empty()) c.error = "read failed";
                const bool last = !ok || c.n == 0;
                const uint64_t t1 = tick();
                rq[i]->push(std::move(c));
                g_times.reader_push += tick() - t1;
                if (last) return;
            }
        });
    }

    // ---- output files (demux.rs:674-688), raw descriptors: whole BGZF members are appended as they come back
    const size_t F = plan.files_per_sample, n_outs = (S + 1) * F;
    {
        rlimit rl;
        if (getrlimit(RLIMIT_NOFILE, &rl) == 0 && rl.rlim_cur < n_outs + 64) {
            rl.rlim_cur = std::min<rlim_t>(rl.rlim_max, n_outs + 64);
            setrlimit(RLIMIT_NOFILE, &rl);
        }
    }
    std::vector<int> fds(n_outs, -1);
    std::vector<std::string> paths(n_outs);
    for (size_t s = 0; s <= S; ++s) {
        const std::string &prefix = s < S ? samples[s].sample_id : opt.unmatched_prefix;
        for (int k = 0; k < 4; ++k) {
            if (!plan.want[k]) continue;
            for (size_t j = 0; j < plan.by_type[k].size(); ++j) {
                const size_t c = s * F + plan.file_base[k] + j;
                paths[c] = opt.output + "/" + prefix + "." + kCodes[k] + std::to_string(j + 1) + ".fq.gz";
                fds[c] = ::open(paths[c].c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
                if (fds[c] < 0) die("cannot create " + paths[c] + ": " + std::strerror(errno));
                std::lock_guard<std::mutex> lk(g_created_mu);
                g_created.push_back(paths[c]);
            }
        }
    }
    info("Created sample and %s writers.", opt.unmatched_prefix.c_str());
    gpu_init.join();
    const double t_ready = now_s();
    for (size_t i = 0; i < n_inputs; ++i)
        for (size_t k = 2; k < kRing; ++k) free_bufs[i]->push(rings[i][k].get());

    // ---- writers: file c belongs to writer c mod W
    // (same box, 64 M templates, tools/ab_writers.sh: one or two writers 24 M templates/s, four 31: 87 k write() calls a second)
    size_t W = std::min<size_t>(4, std::max<size_t>(1, std::min<size_t>(opt.threads, usable_cpus()) / 4));
    if (const char *w = std::getenv("FQTK_WRITERS")) if (*w) W = std::max(1, std::atoi(w));   // (A/B runs)
    std::mutex wmu;
    std::condition_variable wcv_go, wcv_done;
    const fqtk_demux_result *wres = nullptr;
    uint64_t wgen = 0;
    size_t wpending = 0;
    bool wstop = false;
    std::vector<std::thread> writers;
    for (size_t w = 0; w < W; ++w)
        writers.emplace_back([&, w] {
            uint64_t seen = 0;
            for (;;) {
                const fqtk_demux_result *r;
                {
                    std::unique_lock<std::mutex> lk(wmu);
                    wcv_go.wait(lk, [&] { return wstop || wgen != seen; });
                    if (wgen == seen) return;
                    seen = wgen;
                    r = wres;
                }
                const uint64_t t0 = tick();
                for (size_t c = w; c < n_outs; c += W) {
                    const uint64_t lo = r->file_off[c], hi = r->file_off[c + 1];
                    if (hi > lo) write_all(fds[c], r->bytes + lo, (size_t)(hi - lo), paths[c]);
                }
                g_times.comp_write += tick() - t0;
                {
                    std::lock_guard<std::mutex> lk(wmu);
                    --wpending;
                }
                wcv_done.notify_all();
            }
        });
    auto write_result = [&](const fqtk_demux_result &r) {
        if (r.n_blocks == 0) return;
        std::unique_lock<std::mutex> lk(wmu);
        wres = &r;
        wpending = W;
        ++wgen;
        wcv_go.notify_all();
        wcv_done.wait(lk, [&] { return wpending == 0; });
    };

    // ---- collector: chunks in order
    // Every device has its own submit thread, one collector takes the chunks back in order: chunk_dispatch.hpp.
    struct Flight { int dev = 0, slot = 0; uint32_t n = 0; uint64_t first_record = 0; fqtk_demux_result res{}; };
    uint64_t blocks_total = 0, skipped = 0;
    auto chunk_error = [&](const Flight &f, const fqtk_demux_result &r) {
        const uint64_t rec = f.first_record + r.error_template;
        const std::string &path = opt.inputs[std::min<size_t>(r.error_input, n_inputs - 1)];
        char head[4096];
        uint32_t n_bases = 0;
        switch (r.error) {
            case FQTK_DEMUX_ERR_NO_AT: die("Unexpected error parsing FASTQs: expected '@' at record " + std::to_string(rec) + " of " + path);
            case FQTK_DEMUX_ERR_NO_PLUS: die("Unexpected error parsing FASTQs: expected '+' at record " + std::to_string(rec) + " of " + path);
            case FQTK_DEMUX_ERR_QUAL_LEN: die("Unexpected error parsing FASTQs: sequence and quality lengths differ at record " + std::to_string(rec) + " of " + path);
            case FQTK_DEMUX_ERR_TOO_SHORT:
                if (fqtk_demuxer_record_text(demuxers[f.dev], f.slot, r.error_input, r.error_template, head, sizeof head, &n_bases) != FQTK_OK) die(fqtk_last_error());
                die("Read " + std::string(head) + " had too few bases to demux " + std::to_string(n_bases) + " vs. " +
                    std::to_string(plan.rs[r.error_input].min_length()) + " needed in read structure " + plan.rs[r.error_input].to_string() + ".");
            case FQTK_DEMUX_ERR_BARCODE_LEN: die(fqtk_last_error());   // the reference's panic sentence (barcode_matching.rs:95-107)
            case FQTK_DEMUX_ERR_HEADER: {
                if (fqtk_demuxer_record_text(demuxers[f.dev], f.slot, 0, r.error_template, head, sizeof head, &n_bases) != FQTK_OK) die(fqtk_last_error());
                static const char *const kWhat[5] = {"", "Can't handle read name with more than 8 segments: ", "Empty comment in FASTQ header: ",
                                                     "Comment in did not have 4 segments: ", "Malformed comment in FASTQ header: "};
                die(std::string(kWhat[std::min<uint32_t>(r.error_detail, 4)]) + head);
            }
            default: die("internal error: the device did not find four lines per record in a chunk of " + path);
        }
    };
    // ---- this thread: cuts the stream of chunks, chunk k to device k mod G; every device's own thread submits its chunks
    struct Job { size_t n = 0; uint64_t first_record = 0; std::vector<RawChunk> in; std::vector<fqtk_fed_window> win; };
    std::vector<uint64_t> fed_end(n_inputs, 0);   // (collector thread) fed text: where the last chunk collected left each input
    std::atomic<bool> first_submit{false};
    double t_first = 0;
    auto submit_chunk = [&](int g, int slot, uint64_t, Job &j) -> Flight {
        if (fed_mode) {
            const uint64_t th = tick();
            if (fqtk_demuxer_submit_windows(demuxers[g], slot, j.win.data(), (uint32_t)j.n) != FQTK_OK) die(std::string("GPU record pipeline: ") + fqtk_last_error());
            g_times.main_handoff += tick() - th;
            Flight f;
            f.dev = g; f.slot = slot; f.n = (uint32_t)j.n; f.first_record = j.first_record;
            return f;
        }
        std::vector<const uint8_t *> text(n_inputs);
        std::vector<uint64_t> text_len(n_inputs);
        for (size_t i = 0; i < n_inputs; ++i) { text[i] = reinterpret_cast<const uint8_t *>(j.in[i].buf->data); text_len[i] = j.in[i].bytes; }
        const uint64_t th = tick();
        if (fqtk_demuxer_submit(demuxers[g], slot, text.data(), text_len.data(), (uint32_t)j.n) != FQTK_OK) die(std::string("GPU record pipeline: ") + fqtk_last_error());
        // (the chunk counts as submitted only once its text has left the host buffers: they go back to the readers here.  Round 6 tried queueing the
        //  next chunk's copy behind this one before waiting -- a fourth page-locked buffer per input and device -- to close the gap a submit leaves
        //  on the link: 52.4 / 48.6 / 46.8 against 53.6 / 45.4 / 49.6 M templates/s on one box, alternating: the readers are the bound, not the gap)
        if (fqtk_demuxer_text_done(demuxers[g], slot) != FQTK_OK) die(std::string("GPU record pipeline: ") + fqtk_last_error());
        for (size_t i = 0; i < n_inputs; ++i) free_bufs[i]->push(j.in[i].buf);
        g_times.main_handoff += tick() - th;
        Flight f;
        f.dev = g; f.slot = slot; f.n = (uint32_t)j.n; f.first_record = j.first_record;
        return f;
    };
    // The collector waits for the chunk and brings its members home; the RETIRE thread has the writers append them, and only then is the chunk's slot free
    // (its page-locked buffers are what the writers read): the next chunk's wait and device-to-host copy run beside the appends.  Measured on one box,
    // alternating with FQTK_NO_RETIRE_THREAD=1 (one thread does both, as until round 6), 64 M templates: plain 56.6 / 53.7 against 57.0 / 55.8, BGZF 59.4 / 57.5
    // against 58.7 / 57.5, gzip 36.2 / 36.6 against 35.7 / 36.7 M templates/s -- no difference at 16 CPUs: the appends are not what a chunk waits for.
    auto collect_chunk = [&](int g, int slot, uint64_t, Flight &) {   // (the chunk's kernels are through and its members are on their way home)
        const uint64_t t0 = tick();
        if (fqtk_demuxer_collect_begin(demuxers[g], slot) != FQTK_OK) die(std::string("GPU record pipeline: ") + fqtk_last_error());
        g_times.main_gpu_wait += tick() - t0;
    };
    auto retire_chunk = [&](int g, int slot, uint64_t, Flight &f) {
        fqtk_demux_result &r = f.res;
        const uint64_t t0 = tick();
        if (fqtk_demuxer_collect(demuxers[g], slot, &r) != FQTK_OK) die(std::string("GPU record pipeline: ") + fqtk_last_error());
        g_times.main_gpu_wait += tick() - t0;
        if (r.error) chunk_error(f, r);
        if (r.text_end) for (size_t i = 0; i < n_inputs; ++i) fed_end[i] = r.text_end[i];
        write_result(f.res);
        blocks_total += f.res.n_blocks;
        skipped += f.res.n_skipped;
    };
    const bool retire_apart = !env_on("FQTK_NO_RETIRE_THREAD");   // (A/B runs: the collector writes too, as it did)
    ChunkDispatcher<Job, Flight> dispatch(G, FQTK_DEMUX_SLOTS, submit_chunk,
                                          retire_apart ? ChunkDispatcher<Job, Flight>::CollectFn(collect_chunk)
                                                       : ChunkDispatcher<Job, Flight>::CollectFn([&](int g, int slot, uint64_t k, Flight &f) { collect_chunk(g, slot, k, f); retire_chunk(g, slot, k, f); }),
                                          retire_apart ? ChunkDispatcher<Job, Flight>::RetireFn(retire_chunk) : nullptr);
    uint64_t k = 0, records = 0, next_log = 1000000;
    // ---- fed mode: a feeder thread per input hands runs of members to the device; this thread cuts chunks by line counts
    std::mutex fmu;
    std::condition_variable fcv;
    std::vector<uint64_t> lines_fed(n_inputs, 0);
    std::vector<char> fed_done(n_inputs, 0);
    std::string feed_error;
    size_t staged_inputs = 0;   // feeders that have their page-locked staging (they start the device together)
    uint64_t lines_taken = 0;   // per input: 4 x templates submitted
    std::vector<uint64_t> blank_tail(n_inputs, 0);   // lines at an input's end that are no record (see below)
    bool tails_looked_at = false;
    bool feed_stop = false;
    if (fed_mode) {
        // An input is fed again when it is less than two chunks ahead of the chunks cut so far.  A feed is a run of members of up
        // to 32 MB / 256 MB of text: a member takes a wavefront a few milliseconds however many are in flight, and the device
        // holds ~4 800 of them at once, so small feeds leave it idle (16 GB/s of text with 96 MB feeds, 3-4x that when full).
        // (six chunks since round 6; with two the cutter waited 0.9 s of a 64 M-template run's 1.1 for the feeders -- a run of members is three chunks of
        //  templates and takes its input's feeder 15 ms to copy, send and inflate -- and 0.1 s with six or eight: 59.0-59.7 -> 60.7-61.2 M templates/s)
        const uint64_t high_water = 4ull * chunk * (uint64_t)std::max<long>(1, env_num("FQTK_FEED_AHEAD", 6));
        for (size_t i = 0; i < n_inputs; ++i)
            readers.emplace_back([&, i] {
                BgzfFile &bf = *bgzf_in[i];
                fqtk_demuxer *const home = demuxers[home_of[i]];   // the device that inflates this input and keeps its text
                void *pin = nullptr;
                size_t pin_cap = 0;
                std::vector<fqtk_inflate_member> run;
                auto fail = [&](const std::string &e) {
                    std::lock_guard<std::mutex> lk(fmu);
                    if (feed_error.empty()) feed_error = e;
                    fed_done[i] = 1;
                    fcv.notify_all();
                };
                {   // (state of the serial-gzip members of this input: a BGZF file may hold one too)
                    // ---- a serial gzip file: stretches of chunks cut ON THE DEVICE at block starts it finds itself (a lane per bit position),
                    // decoded by a wavefront each without their windows, accepted where the chain of block boundaries proves the parse
                    // (host/parallel_gunzip.hpp's rule), resolved on the device.  A stretch takes the device about as long as its longest chunk
                    // takes ONE wavefront, so chunks are small (64 KiB of file: ~10 ms) and many (1024 a stretch: a quarter of the chip's
                    // wavefronts per input).  Whatever a stretch cannot be cut or decoded like this -- no block start in it, a block that
                    // expands beyond any room, a decoder in doubt -- is decoded by this thread's sequential decoder (region_inflate.hpp) and
                    // handed to the device as text: a VALID file is never refused (demux.rs:844-849 reads any), and a stream is called
                    // corrupt only when the sequential decoder says so too.  (FQTK_TIMING=1 prints every stretch's clock.)
                    static const size_t kChunkBytes = (size_t)std::max<long>(4, std::min<long>(4096, env_num("FQTK_GZ_DEVICE_CHUNK_KB", 64))) << 10;
                    static const size_t kSlotsMax = (size_t)std::max<long>(1, std::min<long>(4096, std::min<long>(env_num("FQTK_GZ_DEVICE_CHUNKS", 2048), (long)((440u << 20) / kChunkBytes))));
                    static const long kForceFallback = env_num("FQTK_GZ_FORCE_FALLBACK", 0);     // (tests: every k-th stretch goes to the host's decoder)
                    static const uint64_t kSymBudget = (uint64_t)std::max<long>(1, env_num("FQTK_GZ_DEVICE_SYM_MB", 1024)) << 20;   // symbols a stretch may ask room for
                    // A stretch takes the device as long as its slowest chunk takes one wavefront -- 20-25 ms whether it has 1023 chunks or 2047 (measured, round 6:
                    // 28 -> 37 M templates/s steady with twice the chunks) -- so stretches are as large as they are useful: about kStretchChunks chunks of TEMPLATES
                    // of this input (a stretch of an index read's file, 60 bytes a record, would otherwise be 58 chunks of templates, decoded long before anyone
                    // asks, in arenas to match), 2048 chunks of file at most; the feeder may be two such stretches ahead of the chunks cut.
                    static const uint64_t kStretchChunks = (uint64_t)std::max<long>(1, env_num("FQTK_GZ_STRETCH_TEMPLATE_CHUNKS", 8));
                    const uint64_t gz_high_water = 4ull * chunk * (uint64_t)env_num("FQTK_GZ_DEVICE_AHEAD", 16);
                    const uint64_t stretch_text_target = kStretchChunks * chunk * (uint64_t)std::max<size_t>(16, per_record_of[i]);
                    auto slots_for = [&](double text_per_byte) {   // chunks of file whose text is the target's
                        const double want = (double)stretch_text_target / (text_per_byte * (double)kChunkBytes);
                        return (size_t)std::max<double>(std::min<double>(want, (double)kSlotsMax), (double)std::min<size_t>(64, kSlotsMax));
                    };
                    const size_t kSlots = std::min(kSlotsMax, std::max<size_t>(slots_for(3.0), std::min<size_t>(64, kSlotsMax)));   // this input's most (its buffers are sized by it): text of 3 : 1 at least
                    size_t slots_now = slots_for(6.0);                       // ... and what a stretch takes: by the ratio measured so far (FASTQ.gz: 4-7 : 1)
                    uint64_t text_seen = 0, bytes_seen = 0;
                    const uint32_t sym_base = (uint32_t)std::max<long>(1, std::min<long>(2048, env_num("FQTK_GZ_DEVICE_SYMS", 8)));
                    uint32_t sym_per_byte = sym_base;   // room per compressed byte: x4 when a chunk runs out, back down by halves after 4 stretches that fit (a run of poly-N
                                                        // reads with constant qualities deflates 1000 : 1 for a megabyte; the rest of the file must not pay for it)
                    size_t stretches_that_fit = 0, fits_at_floor = 0;
                    uint32_t sym_floor = sym_base;     // ... but not below twice the room that last ran out: an input that deflates 15 : 1 throughout (an index read's file through
                                                       // `gzip -6`) would otherwise come back down to 8, lose a stretch, go up again -- every fifth stretch; the floor itself halves after
                                                       // 32 stretches that fit at it, so a burst's room is given back in the end
                    size_t n_stretches = 0, n_chunks_total = 0, n_refused = 0, n_fallbacks = 0;
                    uint64_t fallback_text = 0;
                    size_t &pos = bf.pos;           // byte of the current member's header
                    size_t stretch_at = 0;          // where in `pin` the stretch in hand begins
                    // (the first stretches are short: a quarter of a stretch holds the first chunk of templates, and the record pipeline starts that much sooner)
                    size_t ramp_slots = std::min<size_t>(kSlots, (size_t)std::max<long>(1, env_num("FQTK_GZ_DEVICE_FIRST_CHUNKS", 256)));
                    bool text_only = false;         // every chunk accepted so far decoded 7-bit text only: the search may insist on that (include/fqtk_demux.h)
                    bool high_literals = false;     // ... and once a chunk of this input gave a code to a literal >= 128, it never does again
                    // The NEXT stretch's bytes are copied into a second page-locked buffer while the device decodes this one: a stretch whose
                    // chunks all count ends in its last chunk, so the next one lies in the file from there on (a stretch cut short by a false
                    // start is copied when it is known, as the first one is).
                    void *pin2 = nullptr;
                    size_t pf0 = 0, pf_len = 0;       // the file bytes pin2 holds
                    std::thread prefetcher;
                    struct JoinPrefetch { std::thread &t; void *&p; ~JoinPrefetch() { if (t.joinable()) t.join(); if (p) fqtk_pinned_free(p); } } join_prefetch{prefetcher, pin2};
                    std::unique_ptr<RegionInflate> seq;
                    std::vector<uint8_t> seq_text;
                    std::vector<uint8_t> win_before(32768), win_after(32768);
                    std::vector<fqtk_stream_end> ends(kSlots);
                    bool all_done = false;
                    bool reserved = false;
                    // one gzip member that is no BGZF member, from its header at bf.pos on; false: the run is over (fail() has been called, or the chunks are all cut)
                    auto serial_member = [&]() -> bool {
                        const size_t hl = gzip_header_len(bf.map + pos, bf.size - pos);
                        if (hl == 0) { fail("Unexpected error parsing FASTQs: corrupt gzip stream: " + std::string(pos ? "bad member header" : "not a gzip file") + " in " + bf.path); return false; }
                        if (!reserved && pos == 0) {
                            // the device's buffers for whole stretches, and arenas for the text the feeder may be ahead by plus a stretch's (eight times its
                            // bytes: more only grows them later), made now: a buffer that grows in mid-run is freed first, and that waits for the device
                            reserved = true;
                            const uint64_t stretch_bytes = std::min<uint64_t>(bf.size, (uint64_t)kSlots * kChunkBytes + 131072);
                            const uint64_t ahead_text = (gz_high_water / 4u) * (uint64_t)std::max<size_t>(16, per_record_of[i]);
                            // (an arena holds what the feeder may be ahead by, the stretch just committed and the one to come: the target's text -- 12 : 1 at most
                            //  of a stretch's bytes -- with a quarter to spare; one that turns out too small is made again when it must be, fqtk_demuxer's fed_make_room)
                            const uint64_t stretch_text = std::min<uint64_t>(stretch_bytes * 12u, stretch_text_target + stretch_text_target / 4u);
                            // (two arenas of this size per input are the run's largest allocations -- 22 GB of device memory for four inputs with the factor 2 they had
                            //  until the end of round 5 -- and on a box whose memory a test-suite had just been through allocating them took 0.85 s before the first
                            //  stretch could go: FQTK_TIMING prints it.  Factor 1: 0.4 s there, the same steady rate.)
                            static const uint64_t kArenaFactor = (uint64_t)std::max<long>(1, env_num("FQTK_GZ_ARENA_FACTOR", 1));
                            const uint64_t arena = std::min<uint64_t>((ahead_text + 2u * stretch_text) * kArenaFactor + (64u << 20), (uint64_t)bf.size * 24u + (64u << 20));
                            if (fqtk_demuxer_stream_reserve(home, (uint32_t)i, stretch_bytes + 8, (uint32_t)kSlots, sym_per_byte, kSlots >= 64 ? arena : 0) != FQTK_OK) {
                                fail(std::string("GPU record pipeline: ") + fqtk_last_error());
                                return false;
                            }
                        }
                        uint64_t verified = (uint64_t)(pos + hl) * 8u;   // a block starts here: the member's first
                        bool member_start = true;
                        uint32_t crc_acc = 0;
                        uint64_t size_acc = 0;
                        // what a stretch came to, whoever decoded it
                        auto stretch_done = [&](uint64_t end_bit, bool final_block, uint64_t fed, uint32_t crc, uint64_t n_text, bool *member_done) -> bool {
                            verified = end_bit;
                            member_start = false;
                            crc_acc = (uint32_t)crc32_combine(crc_acc, crc, (z_off_t)n_text);
                            size_acc += n_text;
                            bool last = false;
                            if (final_block) {
                                const size_t trailer = (size_t)((verified + 7u) / 8u);
                                uint32_t want_crc, want_size;
                                std::memcpy(&want_crc, bf.map + trailer, 4);
                                std::memcpy(&want_size, bf.map + trailer + 4, 4);
                                if (want_crc != crc_acc) { fail("Unexpected error parsing FASTQs: corrupt gzip stream: CRC mismatch in " + bf.path); return false; }
                                if (want_size != (uint32_t)size_acc) { fail("Unexpected error parsing FASTQs: corrupt gzip stream: length mismatch in " + bf.path); return false; }
                                *member_done = true;
                                pos = trailer + 8;
                                last = !(pos + 10 <= bf.size && bf.map[pos] == 0x1f && bf.map[pos + 1] == 0x8b);
                                if (last) { all_done = true; bf.pos = bf.size; }
                            }
                            {
                                std::lock_guard<std::mutex> lk(fmu);
                                lines_fed[i] = fed;
                                if (last) fed_done[i] = 1;
                            }
                            fcv.notify_all();
                            return true;
                        };
                        // whether the member ends with the block that ends at end_bit, and the file with the member (trailing garbage is ignored, as zlib's gzread does)
                        auto ends_file = [&](uint64_t end_bit, bool final_block, bool *ok) -> bool {
                            *ok = true;
                            if (!final_block) return false;
                            const size_t trailer = (size_t)((end_bit + 7u) / 8u);
                            if (trailer + 8 > bf.size) { fail("Unexpected error parsing FASTQs: corrupt gzip stream: truncated trailer in " + bf.path); *ok = false; return false; }
                            const size_t nx = trailer + 8;
                            return !(nx + 10 <= bf.size && bf.map[nx] == 0x1f && bf.map[nx + 1] == 0x8b);
                        };
                        // the host's sequential decoder takes the stream from `verified` to the first block boundary at or behind until_bit
                        auto host_stretch = [&](uint64_t until_bit, const char *why, bool *member_done) -> bool {
                            const uint64_t t0 = tick();
                            if (!seq) { seq = std::make_unique<RegionInflate>(); seq->attach(bf.map, bf.size); }
                            if (!member_start && fqtk_demuxer_stream_window(home, (uint32_t)i, win_before.data()) != FQTK_OK) {
                                fail(std::string("GPU record pipeline: ") + fqtk_last_error());
                                return false;
                            }
                            seq_text.clear();
                            uint64_t end_bit = 0;
                            bool final_block = false;
                            std::string e;
                            if (!seq->run(verified, member_start ? nullptr : win_before.data(), until_bit, 256u << 20, &seq_text, &end_bit, &final_block, win_after.data(), &e,
                                          (size_t)std::min<uint64_t>(size_acc, 32768u))) {
                                fail("Unexpected error parsing FASTQs: " + e + " in " + bf.path);
                                return false;
                            }
                            bool ok = true;
                            const bool last = ends_file(end_bit, final_block, &ok);
                            if (!ok) return false;
                            uint64_t fed = 0;
                            uint32_t crc = 0;
                            if (fqtk_demuxer_stream_commit_text(home, (uint32_t)i, seq_text.data(), seq_text.size(), final_block ? nullptr : win_after.data(), last ? 1 : 0, &fed, &crc) != FQTK_OK) {
                                fail(std::string("GPU record pipeline: ") + fqtk_last_error());
                                return false;
                            }
                            ++n_fallbacks;
                            fallback_text += seq_text.size();
                            if (g_timing) info("(timing) gzip input %zu: %zu KB of file decoded by the host's sequential decoder (%s): %zu MB of text in %.0f ms.", i,
                                               (size_t)((end_bit - verified) >> 13), why, seq_text.size() >> 20, (tick() - t0) / 1e6);
                            return stretch_done(end_bit, final_block, fed, crc, seq_text.size(), member_done);
                        };
                        for (bool member_done = false; !member_done;) {
                            {
                                std::unique_lock<std::mutex> lk(fmu);
                                fcv.wait(lk, [&] { return feed_stop || lines_fed[i] < lines_taken + gz_high_water; });
                                if (feed_stop) { return false; }
                            }
                            const uint64_t t0 = tick();
                            // the stretch: from the dword of the verified bit, as many chunks as the symbol budget allows, and a block's worth behind them
                            const size_t b0 = (size_t)(verified / 8u) & ~(size_t)3;
                            size_t n_slots = std::max<size_t>(1, std::min<size_t>(std::min(ramp_slots, slots_now), (size_t)(kSymBudget / ((uint64_t)kChunkBytes * sym_per_byte))));
                            ramp_slots = std::min(kSlots, ramp_slots * 2);
                            const size_t b1 = std::min<size_t>(bf.size, b0 + n_slots * kChunkBytes + 131072);
                            const bool to_end = b1 == bf.size;
                            const size_t bytes = b1 - b0;
                            if (to_end) n_slots = std::max<size_t>(1, std::min(n_slots, (bytes + kChunkBytes - 1) / kChunkBytes));
                            if (bytes + 64 > pin_cap) {   // (once)
                                if (pin) fqtk_pinned_free(pin);
                                pin_cap = std::min<size_t>(bf.size, (kSlots + 5) * kChunkBytes + 131072 + 4) + 65536;
                                if (pinned_alloc_timed(pin_cap, &pin) != FQTK_OK) { pin = nullptr; fail(std::string("cannot allocate page-locked memory: ") + fqtk_last_error()); return false; }
                            }
                            if (kForceFallback > 0 && (n_stretches + n_fallbacks) % (size_t)kForceFallback == (size_t)kForceFallback - 1) {
                                if (!host_stretch(std::min<uint64_t>(verified + (uint64_t)kChunkBytes * 8u * 3u, (uint64_t)bf.size * 8u), "forced", &member_done)) return false;
                                continue;
                            }
                            if (prefetcher.joinable()) prefetcher.join();
                            if (pin2 && pf_len && b0 >= pf0 && b1 <= pf0 + pf_len) {   // the copy made meanwhile holds this stretch
                                std::swap(pin, pin2);
                                stretch_at = b0 - pf0;
                            } else {
                                std::memcpy(pin, bf.map + b0, bytes);
                                stretch_at = 0;
                            }
                            pf_len = 0;
                            if (!to_end) {
                                const size_t from = b1 > 131072 + 4 * kChunkBytes ? (b1 - 131072 - 4 * kChunkBytes) & ~(size_t)3 : 0;
                                const size_t len = std::min<size_t>(bf.size - from, pin_cap - 64);
                                pf0 = from;
                                prefetcher = std::thread([&, from, len] {
                                    if (!pin2 && pinned_alloc_timed(pin_cap, &pin2) != FQTK_OK) { pin2 = nullptr; return; }
                                    std::memcpy(pin2, bf.map + from, len);
                                    pf_len = len;
                                });
                            }
                            const uint8_t *const stretch = static_cast<const uint8_t *>(pin) + stretch_at;
                            const uint64_t tc1 = tick();
                            g_times.reader_parse += tc1 - t0;
                            uint32_t n_chunks = 0;
                            if (fqtk_demuxer_stream_scan(home, (uint32_t)i, stretch, bytes, verified - (uint64_t)b0 * 8u, (uint32_t)kChunkBytes, (uint32_t)n_slots,
                                                         to_end ? 1 : 0, sym_per_byte, text_only ? FQTK_STREAM_SCAN_TEXT : 0u, ends.data(), &n_chunks) != FQTK_OK) {
                                fail(std::string("GPU record pipeline: ") + fqtk_last_error());
                                return false;
                            }
                            const uint64_t t_dec = tick();
                            // a chunk counts if the chunk before it, itself accepted, ended on exactly the bit it started at -- and as far as it
                            // decoded: one that ran out of room or of bytes behind a block boundary ends the stretch at that boundary
                            size_t n_accept = 0;
                            bool out_of_room = false;
                            for (size_t k = 0; k < n_chunks; ++k) {
                                const fqtk_stream_end &e = ends[k];
                                if (k && ends[k - 1].end_bit != e.start_bit) break;
                                if (e.status == 0 && e.end_bit <= (uint64_t)bytes * 8u) {
                                    n_accept = k + 1;
                                    if (e.final_block) break;
                                    continue;
                                }
                                if (e.status == 7) out_of_room = true;
                                if ((e.status == 7 || e.status == 8) && e.n_blocks > 0 && e.end_bit <= (uint64_t)bytes * 8u) n_accept = k + 1;
                                break;
                            }
                            const bool more_room = out_of_room && sym_per_byte < 2048;
                            if (more_room) sym_per_byte = std::min<uint32_t>(2048, sym_per_byte * 4);   // (the next stretches are given more room)
                            if (out_of_room) {
                                const uint32_t failed = more_room ? sym_per_byte / 4u : sym_per_byte;   // (the room the stretch had)
                                sym_floor = std::max(sym_floor, std::min<uint32_t>(2048u, failed * 2u));
                                stretches_that_fit = fits_at_floor = 0;
                            } else if (sym_per_byte > sym_floor) {
                                if (++stretches_that_fit >= 4) { sym_per_byte = std::max(sym_floor, sym_per_byte / 2); stretches_that_fit = 0; }
                            } else if (sym_floor > sym_base && ++fits_at_floor >= 32) {
                                sym_floor = std::max(sym_base, sym_floor / 2);
                                sym_per_byte = std::max(sym_floor, sym_per_byte / 2);
                                fits_at_floor = 0;
                            }
                            if (n_accept == 0) {
                                // chunk 0 starts at a verified boundary and did not get through one block.  Out of room: again with more, while there
                                // is more to give; anything else (no block start in the whole stretch and the block longer than it, a parse error):
                                // the sequential decoder takes this stretch -- if the stream is corrupt, it is the one to say so.
                                if (ends[0].status == 7 && more_room && kChunkBytes * (uint64_t)sym_per_byte <= kSymBudget) { ++n_refused; continue; }
                                static const char *const kWhat[12] = {"", "reserved block type", "stored block length check", "bad code lengths", "over-subscribed or incomplete Huffman code",
                                                                      "invalid code", "distance too far back", "a block that expands beyond the room for symbols", "a block longer than a stretch",
                                                                      "", "", ""};
                                const uint64_t until = n_chunks > 1 ? (uint64_t)b0 * 8u + ends[1].start_bit : (uint64_t)b1 * 8u;
                                if (!host_stretch(until, kWhat[std::min<uint32_t>(ends[0].status, 11)], &member_done)) return false;
                                continue;
                            }
                            for (size_t k = 0; k < n_accept; ++k) if (ends[k].flags & FQTK_STREAM_END_HIGH_LITERALS) high_literals = true;
                            text_only = !high_literals && !env_on("FQTK_GZ_NO_TEXT_FILTER");
                            const fqtk_stream_end &le = ends[n_accept - 1];
                            const uint64_t end_bit = (uint64_t)b0 * 8u + le.end_bit;
                            const bool final_block = le.status == 0 && le.final_block;
                            bool ok = true;
                            const bool last = ends_file(end_bit, final_block, &ok);
                            if (!ok) return false;
                            uint64_t fed = 0, n_text = 0;
                            uint32_t crc = 0;
                            if (fqtk_demuxer_stream_commit(home, (uint32_t)i, (uint32_t)n_accept, member_start ? 1 : 0, last ? 1 : 0, &fed, &crc, &n_text) != FQTK_OK) {
                                fail(std::string("GPU record pipeline: ") + fqtk_last_error());
                                return false;
                            }
                            g_times.reader_push += tick() - tc1;
                            if (g_timing) info("(timing) gzip input %zu: stretch of %u chunks (%zu MB -> %zu MB): copy %.1f ms, search + decode %.1f ms, commit %.1f ms, %zu accepted%s.", i, n_chunks,
                                               bytes >> 20, (size_t)(n_text >> 20), (tc1 - t0) / 1e6, (t_dec - tc1) / 1e6, (tick() - t_dec) / 1e6, n_accept, out_of_room ? " (a chunk ran out of room)" : "");
                            if (g_timing && n_accept < n_chunks && !final_block && n_stretches < 4) {
                                const fqtk_stream_end &a = ends[n_accept - 1], &b = ends[n_accept];
                                info("(timing) gzip input %zu: the chain ends behind chunk %zu (status %u, %u blocks, ended at bit %llu); chunk %zu starts at bit %llu (status %u, %u blocks).", i,
                                     n_accept - 1, a.status, a.n_blocks, (unsigned long long)a.end_bit, n_accept, (unsigned long long)b.start_bit, b.status, b.n_blocks);
                            }
                            ++n_stretches; n_chunks_total += n_chunks; n_refused += n_chunks - n_accept;
                            text_seen += n_text;
                            bytes_seen += (end_bit - verified) / 8u;
                            if (bytes_seen >= (1u << 20)) slots_now = std::min(kSlots, slots_for(std::max(1.0, (double)text_seen / (double)bytes_seen)));
                            if (!stretch_done(end_bit, final_block, fed, crc, n_text, &member_done)) return false;
                            if (b0 > (64u << 20)) madvise(const_cast<uint8_t *>(bf.map), (b0 - (64u << 20)) & ~(size_t)4095, MADV_DONTNEED);
                        }
                        return true;
                    };
                    auto more_members = [&] { return bf.pos + 18 <= bf.size && bf.map[bf.pos] == 0x1f && bf.map[bf.pos + 1] == 0x8b; };
                    void *run_pin = nullptr;           // (runs of BGZF members have their own staging: the stretches' buffers are in use by the prefetcher)
                    size_t run_pin_cap = 0;
                    struct FreeRunPin { void *&p; ~FreeRunPin() { if (p) fqtk_pinned_free(p); } } free_run_pin{run_pin};
                    // A serial gzip input's two staging buffers are page-locked HERE, and no input starts the device before every input has its own: a
                    // page-locking call that runs beside kernels and copies takes 0.1-0.3 s for 64 MB instead of 0.01-0.03 (it waits for the device's work in
                    // flight, and every other HIP call of the process waits for it) -- on a box whose memory a test-suite had just been through, the eight of them
                    // kept a run's FIRST chunk until 1.1-1.3 s and put the run behind the host decoders' (tools/gz_ab.sh after pytest: 2.1-2.3 s against 1.7-1.8).
                    {
                        bool staged_ok = true;
                        if (!BgzfFile::looks_like_bgzf(bf.map + bf.pos, bf.size - bf.pos)) {
                            pin_cap = std::min<size_t>(bf.size, (kSlots + 5) * kChunkBytes + 131072 + 4) + 65536;
                            if (pinned_alloc_timed(pin_cap, &pin) != FQTK_OK) { pin = nullptr; pin_cap = 0; staged_ok = false; }
                            else if (pinned_alloc_timed(pin_cap, &pin2) != FQTK_OK) pin2 = nullptr;   // (then stretches are copied when they are due)
                        } else {   // a BGZF file: the staging of its runs of members (a run is a quarter of FQTK_FEED_TEXT_MB of file at most)
                            const char *v = std::getenv("FQTK_FEED_TEXT_MB");
                            const size_t run_bytes = std::min<size_t>(bf.size - bf.pos, ((size_t)(v && *v ? std::atol(v) : 256) << 20) / 4);
                            run_pin_cap = run_bytes + run_bytes / 4 + 65536;
                            if (pinned_alloc_timed(run_pin_cap, &run_pin) != FQTK_OK) { run_pin = nullptr; run_pin_cap = 0; }   // (made when the first run is due, then)
                        }
                        {
                            std::unique_lock<std::mutex> lk(fmu);
                            ++staged_inputs;
                            fcv.notify_all();
                            fcv.wait(lk, [&] { return staged_inputs >= n_inputs; });
                        }
                        if (!staged_ok) {   // (after the barrier: the other feeders are not left waiting)
                            fail(std::string("cannot allocate page-locked memory: ") + fqtk_last_error());
                            return;
                        }
                    }
                    for (;;) {
                        if (!BgzfFile::looks_like_bgzf(bf.map + bf.pos, bf.size - bf.pos)) {
                            // a gzip member without the BC field (`cat a.bgz b.gz`, or the whole file): one serial stream, decoded in chunks
                            if (!serial_member()) break;
                            if (all_done) break;
                            continue;
                        }
                        {
                            std::unique_lock<std::mutex> lk(fmu);
                            fcv.wait(lk, [&] { return feed_stop || lines_fed[i] < lines_taken + high_water; });
                            if (feed_stop) break;
                        }
                        size_t from = 0, upto = 0;
                        std::string e;
                        const uint64_t t0 = tick();
                        static const size_t run_text = [] { const char *v = std::getenv("FQTK_FEED_TEXT_MB"); return (size_t)(v && *v ? std::atol(v) : 256) << 20; }();
                        if (!bf.next_run(run_text / 4, run_text, &run, &from, &upto, &e)) { fail(e); break; }
                        const size_t bytes = upto - from;
                        if (bytes + 64 > run_pin_cap) {
                            if (run_pin) fqtk_pinned_free(run_pin);
                            run_pin_cap = bytes + bytes / 4 + 65536;
                            if (pinned_alloc_timed(run_pin_cap, &run_pin) != FQTK_OK) { run_pin = nullptr; fail(std::string("cannot allocate page-locked memory: ") + fqtk_last_error()); break; }
                        }
                        if (bytes) std::memcpy(run_pin, bf.map + from, bytes);
                        if (upto > (64u << 20)) madvise(const_cast<uint8_t *>(bf.map), (upto - (64u << 20)) & ~(size_t)4095, MADV_DONTNEED);
                        g_times.reader_parse += tick() - t0;
                        const bool last = !more_members();   // (what is no gzip member behind the last one is ignored, as zlib's gzread and the host path do)
                        uint64_t fed = 0;
                        const uint64_t t1 = tick();
                        if (fqtk_demuxer_feed(home, (uint32_t)i, static_cast<const uint8_t *>(run_pin), bytes, run.data(), (uint32_t)run.size(), last ? 1 : 0, &fed) != FQTK_OK) {
                            fail("Unexpected error parsing FASTQs: " + std::string(fqtk_last_error()) + " in " + bf.path);
                            break;
                        }
                        g_times.reader_push += tick() - t1;
                        if (last) bf.pos = bf.size;
                        {
                            std::lock_guard<std::mutex> lk(fmu);
                            lines_fed[i] = fed;
                            if (last) fed_done[i] = 1;
                        }
                        fcv.notify_all();
                        if (last) break;
                    }
                    if (prefetcher.joinable()) prefetcher.join();
                    if (pin) fqtk_pinned_free(pin);
                    if (g_timing && (n_stretches || n_fallbacks))
                        info("(timing) gzip input %zu: %zu chunks in %zu stretches decoded on the device, %zu not accepted (their stretch was cut there); %zu stretches (%zu MB of text) by the host's sequential decoder.",
                             i, n_chunks_total, n_stretches, n_refused, n_fallbacks, (size_t)(fallback_text >> 20));
                }
            });
        for (;; ++k) {
            size_t n = chunk;
            {
                std::unique_lock<std::mutex> lk(fmu);
                const uint64_t tw = tick();
                fcv.wait(lk, [&] {
                    if (!feed_error.empty()) return true;
                    for (size_t i = 0; i < n_inputs; ++i)
                        if (!fed_done[i] && lines_fed[i] < lines_taken + 4ull * chunk) return false;
                    return true;
                });
                g_times.main_wait += tick() - tw;
                if (!feed_error.empty()) die(feed_error);
                bool all_fed = true;
                for (size_t i = 0; i < n_inputs; ++i) all_fed = all_fed && fed_done[i];
                if (all_fed && !tails_looked_at) {
                    // Every input is in: the host path drops up to three blank lines behind the last record (fastq_io.hpp: next_raw),
                    // and the device has added a newline to every text -- three blank lines and that one would make a record of four
                    // blank lines here.  Such a tail does not count.
                    tails_looked_at = true;
                    for (size_t i = 0; i < n_inputs; ++i) {
                        const uint64_t left_lines = lines_fed[i] - lines_taken;
                        if (left_lines < 4 || left_lines % 4) continue;
                        uint8_t last[16];
                        uint64_t have = 0;
                        if (fqtk_demuxer_fed_tail(demuxers[home_of[i]], (uint32_t)i, ~0ull, last, sizeof last, &have) != FQTK_OK) die(fqtk_last_error());
                        size_t q = (size_t)std::min<uint64_t>(have, sizeof last), newlines = 0;
                        while (q > 0 && (last[q - 1] == '\n' || last[q - 1] == '\r')) newlines += last[--q] == '\n';
                        // (four blank lines: four newlines with nothing but '\r' between them, behind the newline that ends the last record -- or the text's start)
                        if (newlines >= 5 || (newlines == 4 && q == 0 && have <= sizeof last)) blank_tail[i] = 4;
                    }
                }
                for (size_t i = 0; i < n_inputs; ++i) n = std::min<size_t>(n, (size_t)((lines_fed[i] - lines_taken - blank_tail[i]) / 4));
            }
            if (n == 0) break;
            if (!first_submit.exchange(true)) { t_first = now_s(); if (g_timing) info("(timing) first chunk cut; anonymous resident memory %zu MB.", rss_anon_mb()); }
            Job j;
            j.n = n;
            j.first_record = records;
            // the chunk's windows are cut by THIS thread, in chunk order (the devices' threads submit in any order) -- where the next n records of every
            // input lie at its home -- once the chunk's slot is free (chunk_dispatch.hpp: a window pins its input's text until it is submitted)
            dispatch.push(std::move(j), [&](Job &jj) {
                jj.win.resize(n_inputs);
                for (size_t i = 0; i < n_inputs; ++i)
                    if (fqtk_demuxer_fed_cut(demuxers[home_of[i]], (uint32_t)i, (uint32_t)jj.n, &jj.win[i]) != FQTK_OK) die(std::string("GPU record pipeline: ") + fqtk_last_error());
            });
            records += n;
            {
                std::lock_guard<std::mutex> lk(fmu);
                lines_taken += 4ull * n;
            }
            fcv.notify_all();
            while (records >= next_log) { info("demultiplexed %llu records", (unsigned long long)next_log); next_log += 1000000; }
        }
        {
            std::lock_guard<std::mutex> lk(fmu);
            feed_stop = true;
        }
        fcv.notify_all();
    }
    for (; !fed_mode; ++k) {
        Job j;
        j.in.resize(n_inputs);
        for (size_t i = 0; i < n_inputs; ++i) {
            const uint64_t tw = tick();
            j.in[i] = rq[i]->pop();
            g_times.main_wait += tick() - tw;
            if (!j.in[i].error.empty()) die(j.in[i].error);
        }
        const size_t n = j.in[0].n;
        for (size_t i = 0; i < n_inputs; ++i)
            if (j.in[i].n != n) die("FASTQ sources out of sync at records: input " + opt.inputs[j.in[i].n < n ? i : 0] + " ended after a different number of records");
        if (n == 0) break;
        if (!first_submit.exchange(true)) t_first = now_s();
        j.n = n;
        j.first_record = records;
        dispatch.push(std::move(j));   // blocks while G x slots chunks are outstanding
        records += n;
        while (records >= next_log) { info("demultiplexed %llu records", (unsigned long long)next_log); next_log += 1000000; }
    }
    for (auto &t : readers) t.join();
    if (fed_mode && !feed_error.empty()) die(feed_error);
    info("Finished reading input FASTQs.");
    dispatch.finish();
    if (fed_mode) {
        // what lies behind the last record of every input: blank lines (and the newline the device added) -- anything else is
        // a truncated record, or records the other inputs do not have
        std::vector<uint8_t> tail(1u << 20);
        for (size_t i = 0; i < n_inputs; ++i) {
            uint64_t left = 0;
            if (fqtk_demuxer_fed_tail(demuxers[home_of[i]], (uint32_t)i, records ? fed_end[i] : 0, tail.data(), tail.size(), &left) != FQTK_OK) die(fqtk_last_error());
            const size_t have = (size_t)std::min<uint64_t>(left, tail.size());
            bool blank = true;
            size_t lines = 0;
            for (size_t q = 0; q < have; ++q) {
                if (tail[q] == '\n') ++lines;
                else if (tail[q] != '\r') blank = false;
            }
            // (up to three blank lines behind the last record, as the host path allows -- fastq_io.hpp: next_raw --, and the newline the device added)
            if (blank && left <= tail.size() && lines <= 4) continue;
            const bool whole_file_fed = fed_done[i] && bgzf_in[i]->at_end();
            if (lines >= 4 || !whole_file_fed) {
                // records the other inputs do not have: name an input that ended first
                size_t short_one = i == 0 ? (n_inputs > 1 ? 1 : 0) : 0;
                uint64_t fewest = ~0ull;
                for (size_t q = 0; q < n_inputs; ++q)
                    if (q != i && lines_fed[q] < fewest) { fewest = lines_fed[q]; short_one = q; }
                die("FASTQ sources out of sync at records: input " + opt.inputs[short_one] + " ended after a different number of records");
            }
            die("Unexpected error parsing FASTQs: truncated record at end of " + opt.inputs[i]);
        }
        for (size_t g = 0; g < G && g_timing; ++g) {
            double inf_s = 0;
            if (fqtk_demuxer_inflate_seconds(demuxers[g], &inf_s) == FQTK_OK) info("device %d: seconds inflating BGZF members / gzip chunks: %.3f", opt.devices[g], inf_s);
        }
    }
    for (size_t g = 0; g < G; ++g) {   // what is left in the files' open blocks
        fqtk_demux_result r;
        if (fqtk_demuxer_flush(demuxers[g], &r) != FQTK_OK) die(std::string("GPU record pipeline: ") + fqtk_last_error());
        write_result(r);
        blocks_total += r.n_blocks;
    }
    {
        std::lock_guard<std::mutex> lk(wmu);
        wstop = true;
    }
    wcv_go.notify_all();
    for (auto &t : writers) t.join();
    for (size_t c = 0; c < n_outs; ++c) {
        if (fds[c] < 0) continue;
        write_all(fds[c], kBgzfEof, sizeof kBgzfEof, paths[c]);
        if (::close(fds[c]) != 0) die("close failed: " + paths[c]);
    }
    const double t_done = now_s();
    info("Output FASTQ writing complete.");
    info("GPU record pipeline: %llu templates in %llu chunks, %llu BGZF blocks; %.3f s from the first chunk to the last byte written (%.2f M templates/s), devices ready at %.3f s.",
         (unsigned long long)records, (unsigned long long)k, (unsigned long long)blocks_total, t_done - t_first,
         t_done > t_first ? records / (t_done - t_first) / 1e6 : 0.0, t_ready);
    if (g_timing) {
        for (size_t g = 0; g < G; ++g) {
            double st[FQTK_DEMUX_STAGES];
            fqtk_demuxer_stage_seconds(demuxers[g], st);
            std::string line;
            for (int q = 0; q < FQTK_DEMUX_STAGES; ++q) line += std::string(q ? ", " : "") + fqtk_demuxer_stage_name(q) + " " + std::to_string(st[q]);
            info("device %d stage seconds: %s", opt.devices[g], line.c_str());
        }
        info("host thread-seconds: readers fill %.2f push %.2f | this thread: waiting for readers %.2f, submit + text copy %.2f | collector waiting for the GPU %.2f | writers %.2f",
             g_times.reader_parse / 1e9, g_times.reader_push / 1e9, g_times.main_wait / 1e9, g_times.main_handoff / 1e9, g_times.main_gpu_wait / 1e9, g_times.comp_write / 1e9);
    }
    if (skipped == 0) info("No records were skipped.");
    else info("%llu records were skipped due to Too few bases", (unsigned long long)skipped);
    report_footprint(n_outs);

    // ---- metrics (demux.rs:994-998): the per-sample counts are a column of the device's placement sums
    std::vector<uint64_t> counts(S + 1, 0);
    for (size_t g = 0; g < G; ++g)
        if (fqtk_demuxer_counts(demuxers[g], counts.data()) != FQTK_OK) die(fqtk_last_error());
    uint64_t sum = 0;
    for (uint64_t c : counts) sum += c;
    if (sum != records - skipped) die("internal error: device counts do not add up to the number of templates");
    std::vector<DemuxMetric> rows(S);
    for (size_t s = 0; s < S; ++s) {
        rows[s].sample_id = samples[s].sample_id;
        rows[s].barcode = samples[s].barcode;
        rows[s].templates = counts[s];
    }
    DemuxMetric unmatched;
    unmatched.sample_id = opt.unmatched_prefix;
    unmatched.barcode = ".";
    unmatched.templates = counts[S];
    update_metrics(rows, unmatched);
    rows.push_back(unmatched);
    std::string err;
    if (!write_metrics_tsv(opt.output + "/demux-metrics.txt", rows, &err)) die(err);
    end_process();
}

}  // namespace

int main(int argc, char **argv) {
    now_s();
    if (!env_on("FQTK_FOREGROUND") && !env_on("FQTK_CLEAN_EXIT") && argc >= 2 && std::string(argv[1]) == "demux") {
        int fds[2];
        if (::pipe(fds) == 0) {
            const pid_t pid = ::fork();
            if (pid > 0) { ::close(fds[1]); supervise(pid, fds[0]); }
            if (pid == 0) {
                ::close(fds[0]);
                ::fcntl(fds[1], F_SETFD, FD_CLOEXEC);
                g_done_fd = fds[1];
                ::prctl(PR_SET_PDEATHSIG, SIGTERM);   // (a parent that is killed outright takes the run with it)
                if (::getppid() == 1) std::_Exit(1);   // (... also one that was gone before the line above)
            } else { ::close(fds[0]); ::close(fds[1]); }   // (no child to be had: the run happens here)
        }
    }
    // Batches of decoded input and blocks of output are tens of MB each and come and go all the time: by default
    // glibc maps and unmaps every one of them, and every fresh page is a fault plus 4 KB of zeroes (measured on
    // gzip inputs: a third of the reader threads' time).  Keep them on the heap, where a freed block is reused.
    if (!std::getenv("FQTK_MALLOC_DEFAULT")) {
        mallopt(M_MMAP_THRESHOLD, 1 << 30);
        mallopt(M_TRIM_THRESHOLD, 1 << 30);
    }
    g_timing = std::getenv("FQTK_TIMING") != nullptr;
    if (argc < 2 || std::string(argv[1]) == "--help" || std::string(argv[1]) == "-h") {
        std::fputs("fqtk (MI355X-native demux)\n\nUsage: fqtk <COMMAND>\n\nCommands:\n  demux  Performs sample demultiplexing on FASTQs\n", stdout);
        return argc < 2 ? 2 : 0;
    }
    if (std::string(argv[1]) != "demux") die(std::string("unrecognized subcommand '") + argv[1] + "' (only `demux` is in scope, see DESIGN.md)");
    Options opt = parse_args(argc - 2, argv + 2);

    // ---- read structures are parsed at argument time by clap in the reference ---------------------
    Plan plan;
    for (const std::string &t : opt.read_structures) {
        ReadStructure r;
        std::string err;
        if (!ReadStructure::parse(t, &r, &err)) die("invalid value '" + t + "' for '--read-structures <READ_STRUCTURES>...': " + err);
        plan.rs.push_back(r);
    }
    bool skip_few = false;
    for (const std::string &s : opt.skip_reasons) {   // demux.rs:69-77
        if (s == "too few bases" || s == "too-few-bases" || s == "toofewbases") skip_few = true;
        else die("invalid value '" + s + "' for '--skip-reasons <SKIP_REASONS>': Invalid skip reason: " + s);
    }

    // ---- validate_and_prepare_inputs (demux.rs:806-875): all problems reported together ----------
    std::vector<std::string> problems;
    if (opt.inputs.size() != plan.rs.size())
        problems.push_back("The same number of read structures should be given as FASTQs " +
                           std::to_string(plan.rs.size()) + " read-structures provided for " +
                           std::to_string(opt.inputs.size()) + " FASTQs");
    struct stat st;
    if (stat(opt.output.c_str(), &st) != 0) {
        info("Output directory \"%s\" didn't exist, creating it.", opt.output.c_str());
        std::string cmd;
        // mkdir -p
        for (size_t p = 1; p <= opt.output.size(); ++p)
            if (p == opt.output.size() || opt.output[p] == '/') {
                std::string sub = opt.output.substr(0, p);
                if (!sub.empty() && stat(sub.c_str(), &st) != 0 && mkdir(sub.c_str(), 0777) != 0) die("cannot create " + sub);
            }
    }
    if (stat(opt.output.c_str(), &st) == 0 && (st.st_mode & 0222) == 0)
        problems.push_back("Ouput directory \"" + opt.output + "\" cannot be read-only");
    for (char c : opt.output_types) {
        SegType t;
        if (!seg_type_from_char(c, &t)) problems.push_back(std::string("Error parsing segment types to report: Read structure had unknown type: ") + c);
    }
    for (const std::string &in : opt.inputs)
        if (access(in.c_str(), F_OK) != 0) problems.push_back("Provided input file \"" + in + "\" doesn't exist");
    std::vector<std::unique_ptr<FastqSource>> sources;
    // Single-stream gzip inputs are decoded by several threads each (parallel_gunzip.hpp): three quarters of the usable
    // CPUs, shared out by file size (the index reads of a run are a fifth of its bytes), at most 8 per file.
    std::vector<unsigned> gz_threads(opt.inputs.size(), 1);
    {
        std::vector<uint64_t> sz(opt.inputs.size(), 0);
        uint64_t total = 0;
        for (size_t i = 0; i < opt.inputs.size(); ++i) {
            struct stat st;
            if (stat(opt.inputs[i].c_str(), &st) == 0 && S_ISREG(st.st_mode)) sz[i] = (uint64_t)st.st_size;
            total += sz[i];
        }
        const unsigned pool = std::max(2u, usable_cpus() * 3 / 4);
        for (size_t i = 0; i < opt.inputs.size(); ++i)
            if (total) gz_threads[i] = (unsigned)std::min<uint64_t>(8, std::max<uint64_t>(1, (pool * sz[i] + total / 2) / total));
    }
    for (size_t i_in = 0; i_in < opt.inputs.size(); ++i_in) {
        const std::string &in = opt.inputs[i_in];
        auto src = std::make_unique<FastqSource>();
        std::string err;
        // (BGZF inputs: the same share of the CPUs as helpers that inflate blocks side by side; unused for other kinds)
        if (access(in.c_str(), F_OK) == 0 && !src->open(in, &err, std::max(2u, gz_threads[i_in]), gz_threads[i_in])) problems.push_back("Error opening input files for reading: " + err);
        sources.push_back(std::move(src));
    }
    if (opt.threads < 5) problems.push_back("Threads provided " + std::to_string(opt.threads) + " was too low! Must be 5 or more.");
    if (problems.empty()) {
        for (char c : opt.output_types)
            for (int k = 0; k < 4; ++k) if ((char)kTypes[k] == c) plan.want[k] = true;
        if (opt.output_types.empty()) problems.push_back("No output types requested, must request at least one output segment type.");
    }
    if (!problems.empty()) {
        std::string details = "Inputs failed validation!\n";
        for (const std::string &p : problems) details += "    - " + p + "\n";
        die("The following errors with the input(s) were detected:\n" + details);
    }

    // ---- samples (demux.rs:884) ----------------------------------------------------------------------
    std::vector<Sample> samples;
    {
        std::string err;
        if (!load_samples(opt.sample_metadata, &samples, &err)) die(err);
    }
    info("%zu samples loaded from file \"%s\"", samples.size(), opt.sample_metadata.c_str());
    if (opt.max_mismatches > 255 || opt.min_mismatch_delta > 255) die("out of range integral type conversion attempted");   // u8::try_from, demux.rs:923-924
    if (opt.compression_level > 255) die("out of range integral type conversion attempted");
    if (opt.compression_level > 12) die("compression level must be at most 12");

    // ---- output plan (demux.rs:660-743): per sample, per requested type, one file per segment -----
    const size_t n_inputs = plan.rs.size();
    for (uint32_t i = 0; i < n_inputs; ++i)
        for (uint32_t s = 0; s < plan.rs[i].segments.size(); ++s)
            for (int k = 0; k < 4; ++k)
                if (plan.rs[i].segments[s].kind == kTypes[k]) plan.by_type[k].push_back({i, s});
    for (int k = 0; k < 4; ++k) {
        plan.file_base[k] = plan.files_per_sample;
        if (plan.want[k]) plan.files_per_sample += plan.by_type[k].size();
    }
    // ---- the matcher (demux.rs:921-926), use_cache = true as the reference passes -----------------
    // Bringing the GPU up (HIP runtime, code objects, table upload, memo build) takes about a second: it runs
    // on its own thread while the readers already decompress and parse the first chunks and this thread
    // creates the output files.
    const size_t S = samples.size();
    std::vector<const char *> bc;
    for (const Sample &s : samples) bc.push_back(s.barcode.c_str());
    // One matcher (replicated table) per device; chunk k is matched on device k mod G.  Templates are
    // independent, so there is no data-path exchange; the per-device counts are reduced at the end.
    if (opt.devices.empty()) opt.devices.push_back(opt.device);
    {
        // Room for every output file in the descriptor table NOW, while this is the only thread: the kernel grows the
        // table of a multi-threaded process behind an RCU grace period per doubling (expand_fdtable), which on these
        // hosts made creating 771 files next to the device bring-up take 0.5 s instead of 2 ms -- and held the HIP
        // runtime's own open() calls up as long.
        const size_t want = (samples.size() + 1) * plan.files_per_sample + 256;
        rlimit rl;
        if (getrlimit(RLIMIT_NOFILE, &rl) == 0) {
            if (rl.rlim_cur < want) { rl.rlim_cur = std::min<rlim_t>(rl.rlim_max, want); setrlimit(RLIMIT_NOFILE, &rl); }
            const int top = (int)std::min<rlim_t>(rl.rlim_cur, want) - 1;
            if (top > 2 && dup2(2, top) == top) ::close(top);
        }
    }
    if (!opt.host_output && !env_on("FQTK_HOST_OUTPUT")) {
        std::string why;
        if (gpu_output_supported(plan, &why)) run_gpu_output(opt, plan, samples, sources, skip_few);   // does not return
        info("The GPU record pipeline does not take this configuration (%s): records are formatted and compressed on the host.", why.c_str());
    }
    const size_t G = opt.devices.size();
    std::vector<fqtk_matcher *> matchers(G, nullptr);
    const uint32_t L = (uint32_t)samples[0].barcode.size();
    std::thread gpu_init([&] {
        for (size_t g = 0; g < G; ++g) {
            if (fqtk_matcher_create(bc.data(), (uint32_t)S, L, (uint8_t)opt.max_mismatches, (uint8_t)opt.min_mismatch_delta,
                                    opt.devices[g], &matchers[g]) != FQTK_OK)
                die(std::string("cannot create the GPU barcode matcher: ") + fqtk_last_error());
            std::vector<const char *> ids;   // so that a length error names the sample like the reference's panic does
            for (const Sample &s : samples) ids.push_back(s.sample_id.c_str());
            fqtk_matcher_set_sample_ids(matchers[g], ids.data());
            info("GPU barcode matcher ready on device %d (%llu memo entries).", opt.devices[g],
                 (unsigned long long)fqtk_matcher_memo_entries(matchers[g]));
        }
    });

    // sample-barcode layout of one template: fixed total length, or variable when a B segment is '+'
    bool variable_barcode = false;
    size_t fixed_barcode_len = 0;
    for (const SegRef &r : plan.by_type[1]) {
        const ReadSegment &seg = plan.rs[r.input].segments[r.seg];
        if (seg.has_length()) fixed_barcode_len += (size_t)seg.length; else variable_barcode = true;
    }

    // Per input: where its fixed-length sample-barcode segments sit inside a read and inside the packed row.
    // Reader threads pack them (and flag too-short reads) while the record is still hot in their cache;
    // the main thread then only interleaves one small fixed-width copy per barcode-carrying input.
    struct InputPack { std::vector<std::pair<size_t, size_t>> segs; size_t width = 0, col = 0, min_len = 0; };
    std::vector<InputPack> ipack(n_inputs);
    {
        size_t col = 0;
        for (const SegRef &r : plan.by_type[1]) {
            const ReadSegment &seg = plan.rs[r.input].segments[r.seg];
            if (!seg.has_length()) continue;
            if (ipack[r.input].width == 0) ipack[r.input].col = col;
            ipack[r.input].segs.emplace_back(seg.offset, (size_t)seg.length);
            ipack[r.input].width += (size_t)seg.length;
            col += (size_t)seg.length;
        }
        for (size_t i = 0; i < n_inputs; ++i) ipack[i].min_len = plan.rs[i].min_length();
    }

    // ---- stage A: one reader thread per input ------------------------------------------------------
    const size_t chunk_reads = std::max<unsigned long>(1, opt.chunk_reads);
    std::vector<std::unique_ptr<BoundedQueue<ReadResult>>> rq;
    std::vector<std::thread> readers;
    for (size_t i = 0; i < n_inputs; ++i) rq.push_back(std::make_unique<BoundedQueue<ReadResult>>(3));
    for (size_t i = 0; i < n_inputs; ++i)
        readers.emplace_back([&, i] {
            for (;;) {
                ReadResult r;
                r.batch = std::make_unique<RecBatch>();
                const uint64_t t0 = tick();
                const bool ok = sources[i]->next_batch(chunk_reads, r.batch.get(), &r.error);
                if (ok) {   // too-few-bases flags + this input's slice of the packed sample-barcode rows
                    RecBatch &b = *r.batch;
                    const InputPack &ip = ipack[i];
                    const size_t n = b.recs.size();
                    b.n_short = 0;
                    b.too_short.assign(n, 0);
                    if (!variable_barcode && ip.width) b.bc.resize(n * ip.width);
                    for (size_t j = 0; j < n; ++j) {
                        if (b.recs[j].seq_len < ip.min_len) { b.too_short[j] = 1; ++b.n_short; continue; }
                        if (variable_barcode || !ip.width) continue;
                        uint8_t *dst = b.bc.data() + j * ip.width;
                        const char *seq = b.seq(j);
                        for (const auto &sg : ip.segs) { std::memcpy(dst, seq + sg.first, sg.second); dst += sg.second; }
                    }
                }
                const uint64_t t1 = tick();
                g_times.reader_parse += t1 - t0;
                if (!ok) {
                    rq[i]->push(std::move(r));
                    return;
                }
                const bool last = r.batch->recs.empty();
                rq[i]->push(std::move(r));
                g_times.reader_push += tick() - t1;
                if (last) return;
            }
        });

    const size_t n_outs = (S + 1) * plan.files_per_sample;
    {
        rlimit rl;
        if (getrlimit(RLIMIT_NOFILE, &rl) == 0 && rl.rlim_cur < n_outs + 64) {
            rl.rlim_cur = std::min<rlim_t>(rl.rlim_max, n_outs + 64);
            setrlimit(RLIMIT_NOFILE, &rl);
        }
    }
    std::vector<OutFile> outs(n_outs);
    for (size_t s = 0; s <= S; ++s) {
        const std::string &prefix = s < S ? samples[s].sample_id : opt.unmatched_prefix;
        for (int k = 0; k < 4; ++k) {
            if (!plan.want[k]) continue;
            for (size_t j = 0; j < plan.by_type[k].size(); ++j) {
                OutFile &of = outs[s * plan.files_per_sample + plan.file_base[k] + j];
                of.path = opt.output + "/" + prefix + "." + kCodes[k] + std::to_string(j + 1) + ".fq.gz";
                of.f = std::fopen(of.path.c_str(), "wb");
                if (!of.f) die("cannot create " + of.path + ": " + std::strerror(errno));
                std::lock_guard<std::mutex> lk(g_created_mu);
                g_created.push_back(of.path);
            }
        }
    }
    info("Created sample and %s writers.", opt.unmatched_prefix.c_str());

    gpu_init.join();

    // ---- stage C: routers (partitioned by sample: one owner per file, input order kept) format the
    //      records; a shared pool BGZF-compresses the 64 KiB blocks and writes them in order ---------
    // (--threads beyond the CPUs the process may use -- affinity mask, cgroup quota -- only adds contention:
    //  measured on a 16-CPU quota, 32 threads ran 18 % slower than 16)
    const size_t n_threads_c = std::min<size_t>(std::max<size_t>(2, opt.threads - 1), std::max<size_t>(4, usable_cpus()));
    // Formatting a template costs ~1.15 us of router time, compressing its ~680 bytes at level 5 ~5.2 us of
    // libdeflate time (measured, FQTK_TIMING, 16 M dual-index templates): two routers feed seven compressors.
    const size_t n_workers = std::max<size_t>(1, (n_threads_c * 2 + 4) / 9);   // routers
    const size_t n_comp = std::max<size_t>(1, n_threads_c - n_workers);     // compressors / writers
    // Output files fill in lock-step (samples are hit in proportion, so hundreds of files reach a full
    // 64 KiB block within the same few chunks): the queue must absorb such a burst or the routers stall
    // on it while the compressors idle between bursts (measured: 1 push in 26 found a 73-deep queue full
    // and waited 4 ms).
    JobQueues jobs;
    for (size_t c = 0; c < n_comp; ++c)
        jobs.push_back(std::make_unique<BoundedQueue<CompressJob>>(std::max<size_t>(64, 8192 / n_comp)));   // <= 512 MiB of blocks in flight
    std::vector<std::thread> compressors;
    for (size_t c = 0; c < n_comp; ++c)
        compressors.emplace_back([&, c] {
            BlockCompressor bc((int)opt.compression_level);
            for (;;) {
                const uint64_t t0 = tick();
                CompressJob j = jobs[c]->pop();
                g_times.comp_wait += tick() - t0;
                if (!j.of) break;
                compress_and_write(j, bc);
                g_pool.put(std::move(j.data));
            }
        });
    std::vector<std::unique_ptr<BoundedQueue<std::shared_ptr<Chunk>>>> wq;
    for (size_t w = 0; w < n_workers; ++w) wq.push_back(std::make_unique<BoundedQueue<std::shared_ptr<Chunk>>>(8));
    // Which router owns which output file.  One owner per file keeps every file in input order without locks;
    // WHICH owner is decided when the first chunk comes back from the GPU, by the per-sample counts of that
    // chunk (longest-processing-time-first over the files): samples are far from equally popular -- the
    // unmatched pair alone takes 10-20 % of the records -- and a modulo assignment left one router with twice the
    // work of the others.  Written once by the main thread before chunk 0 is handed to the routers.
    std::vector<uint16_t> file_owner(n_outs, 0);
    std::vector<std::thread> workers;
    for (size_t w = 0; w < n_workers; ++w)
        workers.emplace_back([&, w] {
            struct Owned { int k; uint32_t j; };
            std::vector<std::vector<Owned>> owned(S + 1);
            bool have_plan = false;
            std::vector<std::string_view> bsegs, msegs;
            std::vector<uint32_t> mine;
            for (;;) {
                const uint64_t tw = tick();
                std::shared_ptr<Chunk> ch = wq[w]->pop();
                const uint64_t tf = tick();
                g_times.router_wait += tf - tw;
                if (ch && !have_plan) {   // file_owner is final once a chunk has been queued
                    have_plan = true;
                    for (size_t s = 0; s <= S; ++s)
                        for (int k = 0; k < 4; ++k) {
                            if (!plan.want[k]) continue;
                            for (uint32_t j = 0; j < plan.by_type[k].size(); ++j)
                                if (file_owner[s * plan.files_per_sample + plan.file_base[k] + j] == w) owned[s].push_back({k, j});
                        }
                }
                if (!ch) break;
                uint64_t t_sub = 0;
                const RecBatch &b0 = *ch->batches[0];   // header of the FIRST input (combine_readsets, demux.rs:126-139)
                // This router's templates of the chunk, in input order.  Their records were last touched by the
                // reader threads on other cores and a chunk (~100 MB) outlives the caches: each record would start
                // with a miss to DRAM.  Knowing the list up front, the lines of the record a few templates ahead
                // are requested while this one is formatted.
                mine.clear();
                for (size_t i = 0; i < ch->n; ++i) {
                    if (ch->skip[i]) continue;
                    const size_t s = ch->res[i].idx == FQTK_NO_MATCH ? S : ch->res[i].idx;
                    if (!owned[s].empty()) mine.push_back((uint32_t)i);
                }
                auto prefetch_template = [&](size_t i) {
                    for (size_t b = 0; b < ch->batches.size(); ++b) {
                        const RecBatch &rb = *ch->batches[b];
                        const FastqRec &r = rb.recs[i];
                        const char *base = rb.rec_base(i);
                        __builtin_prefetch(base + r.head_off);
                        for (uint32_t off = 0; off < r.seq_len; off += 64) {
                            __builtin_prefetch(base + r.seq_off + off);
                            __builtin_prefetch(base + r.qual_off + off);
                        }
                    }
                };
                constexpr size_t kAhead = 6;
                for (size_t q = 0; q < std::min(kAhead, mine.size()); ++q) prefetch_template(mine[q]);
                for (size_t q = 0; q < mine.size(); ++q) {
                    if (q + kAhead < mine.size()) prefetch_template(mine[q + kAhead]);
                    const size_t i = mine[q];
                    const size_t s = ch->res[i].idx == FQTK_NO_MATCH ? S : ch->res[i].idx;
                    auto span = [&](const SegRef &r, std::string_view *bases, std::string_view *quals) {
                        const RecBatch &b = *ch->batches[r.input];
                        size_t lo, hi;
                        segment_span(plan.rs[r.input].segments[r.seg], b.recs[i].seq_len, &lo, &hi);
                        *bases = std::string_view(b.seq(i) + lo, hi - lo);
                        *quals = std::string_view(b.qual(i) + lo, hi - lo);
                    };
                    bsegs.clear();
                    msegs.clear();
                    std::string_view sv, qv;
                    for (const SegRef &r : plan.by_type[1]) { span(r, &sv, &qv); bsegs.push_back(sv); }
                    for (const SegRef &r : plan.by_type[2]) { span(r, &sv, &qv); msegs.push_back(sv); }
                    const std::string_view header(b0.head(i), b0.recs[i].head_len);
                    for (const Owned &o : owned[s]) {
                        OutFile &of = outs[s * plan.files_per_sample + plan.file_base[o.k] + o.j];
                        std::string err;
                        if (!write_header(of.buf, o.j + 1, header, bsegs, msegs, &err)) die(err);
                        span(plan.by_type[o.k][o.j], &sv, &qv);
                        of.buf.push_back('\n');
                        of.buf.append(sv);
                        of.buf.append("\n+\n");
                        of.buf.append(qv);
                        of.buf.push_back('\n');
                        if (of.buf.size() >= kBgzfBlockSize) {
                            const uint64_t ts = tick();
                            submit_blocks(of, jobs, false);
                            t_sub += tick() - ts;
                        }
                    }
                }
                g_times.router_submit += t_sub;
                g_times.router_format += tick() - tf - t_sub;
            }
            for (size_t f = 0; f < outs.size(); ++f)
                if (file_owner[f] == w) submit_blocks(outs[f], jobs, true);
        });

    // ---- stage B (this thread): chunk assembly, barcode SoA packing, GPU pipeline -------------------
    const int kSlots = 2 * (int)G;   // two pipeline slots per device; global slot gs -> device gs % G, local slot gs / G
    struct SlotBuf { uint8_t *obs = nullptr; uint32_t *lens = nullptr; fqtk_match_t *out = nullptr; size_t obs_cap = 0, n_cap = 0; };
    std::vector<SlotBuf> sb(kSlots);
    auto ensure_slot = [&](SlotBuf &b, size_t n, size_t stride) {
        if (n * stride > b.obs_cap) {
            if (b.obs) fqtk_pinned_free(b.obs);
            void *p = nullptr;
            b.obs_cap = n * stride + (n * stride) / 4 + 64;
            if (pinned_alloc_timed(b.obs_cap, &p) != FQTK_OK) die(fqtk_last_error());
            b.obs = (uint8_t *)p;
        }
        if (n > b.n_cap) {
            if (b.out) fqtk_pinned_free(b.out);
            if (b.lens) fqtk_pinned_free(b.lens);
            void *p = nullptr, *q = nullptr;
            b.n_cap = n + n / 4 + 16;
            if (pinned_alloc_timed(b.n_cap * sizeof(fqtk_match_t), &p) != FQTK_OK) die(fqtk_last_error());
            if (pinned_alloc_timed(b.n_cap * sizeof(uint32_t), &q) != FQTK_OK) die(fqtk_last_error());
            b.out = (fqtk_match_t *)p;
            b.lens = (uint32_t *)q;
        }
    };
    struct Pending { std::shared_ptr<Chunk> chunk; std::vector<uint32_t> rows; int slot = -1; bool identity = false; size_t n_rows = 0; };
    bool owners_assigned = false;
    auto finish = [&](Pending &p) {
        if (!p.chunk) return;
        if (p.slot >= 0) {
            const uint64_t tg = tick();
            if (fqtk_matcher_wait(matchers[p.slot % G], p.slot / (int)G) != FQTK_OK)
                die(std::string(fqtk_last_error()));   // over-long barcode: the reference panics too (barcode_matching.rs:95-107)
            g_times.main_gpu_wait += tick() - tg;
            if (p.identity) std::memcpy(p.chunk->res.data(), sb[p.slot].out, p.n_rows * sizeof(fqtk_match_t));   // no skipped template
            else for (size_t j = 0; j < p.rows.size(); ++j) p.chunk->res[p.rows[j]] = sb[p.slot].out[j];
        }
        if (!owners_assigned) {   // first chunk back from the GPU: balance the files over the routers by its counts
            owners_assigned = true;
            std::vector<uint64_t> cnt(S + 1, 0);
            for (size_t j = 0; j < p.chunk->n; ++j)
                if (!p.chunk->skip[j]) ++cnt[p.chunk->res[j].idx == FQTK_NO_MATCH ? S : p.chunk->res[j].idx];
            std::vector<size_t> order(n_outs);
            for (size_t f = 0; f < n_outs; ++f) order[f] = f;
            auto weight = [&](size_t f) { return cnt[f / plan.files_per_sample]; };
            std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return weight(a) > weight(b); });
            std::vector<uint64_t> load(n_workers, 0);
            size_t rr = 0;
            for (size_t f : order) {
                size_t best = 0;
                if (weight(f) == 0) best = rr++ % n_workers;   // nothing seen yet: spread evenly
                else for (size_t w = 1; w < n_workers; ++w) if (load[w] < load[best]) best = w;
                file_owner[f] = (uint16_t)best;
                load[best] += weight(f) + 1;
            }
        }
        const uint64_t th = tick();
        for (size_t w = 0; w < n_workers; ++w) wq[w]->push(p.chunk);
        g_times.main_handoff += tick() - th;
        p.chunk.reset();
    };
    std::vector<Pending> pending(kSlots);
    uint64_t total_templates = 0, skipped = 0, k = 0, next_log = 1000000;
    for (;; ++k) {
        auto ch = std::make_shared<Chunk>();
        size_t n_nonempty = 0;
        for (size_t i = 0; i < n_inputs; ++i) {
            const uint64_t tw = tick();
            ReadResult r = rq[i]->pop();
            g_times.main_wait += tick() - tw;
            if (!r.error.empty()) die(r.error);
            if (!r.batch->recs.empty()) ++n_nonempty;
            ch->batches.push_back(std::move(r.batch));
        }
        if (n_nonempty == 0) break;
        ch->n = ch->batches[0]->recs.size();
        for (size_t i = 0; i < n_inputs; ++i)
            if (ch->batches[i]->recs.size() != ch->n)
                die("FASTQ sources out of sync at records: input " + opt.inputs[i] + " ended after a different number of records");
        ch->skip.assign(ch->n, 0);
        ch->res.assign(ch->n, fqtk_match_t{FQTK_NO_MATCH, 255, 255});
        // too-few-bases rule (demux.rs:298-313): skip the whole template, or fail with the reference's
        // text for the FIRST offending template (templates in order, inputs in order within one).
        // The readers flagged the short reads; the common chunk has none.
        size_t n_skip = 0;
        {
            size_t bad_j = ch->n, bad_i = 0;
            for (size_t i = 0; i < n_inputs; ++i) {
                const RecBatch &b = *ch->batches[i];
                if (b.n_short == 0) continue;
                for (size_t j = 0; j < ch->n; ++j)
                    if (b.too_short[j]) {
                        if (!ch->skip[j]) { ch->skip[j] = 1; ++n_skip; }
                        if (j < bad_j) { bad_j = j; bad_i = i; }
                    }
            }
            if (bad_j < ch->n && !skip_few) {
                const RecBatch &b = *ch->batches[bad_i];
                die("Read " + std::string(b.head(bad_j), b.recs[bad_j].head_len) + " had too few bases to demux " +
                    std::to_string(b.recs[bad_j].seq_len) + " vs. " + std::to_string(plan.rs[bad_i].min_length()) +
                    " needed in read structure " + plan.rs[bad_i].to_string() + ".");
            }
        }
        const int slot = (int)(k % kSlots);
        finish(pending[slot]);   // the chunk that used this slot two iterations ago
        Pending &p = pending[slot];
        p.chunk = ch;
        p.rows.clear();
        p.slot = -1;
        p.identity = n_skip == 0;
        // pack the sample barcodes: concatenation of all B segments in input order (demux.rs:121-123)
        size_t stride = (fixed_barcode_len + 3) / 4 * 4;
        if (variable_barcode) {
            size_t mx = fixed_barcode_len;
            for (size_t j = 0; j < ch->n; ++j) {
                if (ch->skip[j]) continue;
                size_t len = 0;
                for (const SegRef &r : plan.by_type[1]) {
                    size_t lo, hi;
                    segment_span(plan.rs[r.input].segments[r.seg], ch->batches[r.input]->recs[j].seq_len, &lo, &hi);
                    len += hi - lo;
                }
                mx = std::max(mx, len);
            }
            stride = (mx + 3) / 4 * 4;
        }
        if (stride == 0) stride = 4;
        ensure_slot(sb[slot], ch->n, stride);
        size_t row = 0;
        if (!variable_barcode) {
            // fixed layout: every row is the readers' pre-packed slices side by side, pad bytes zero
            uint8_t *base = sb[slot].obs;
            if (stride != fixed_barcode_len) std::memset(base, 0, (ch->n - n_skip) * stride);
            for (size_t i = 0; i < n_inputs; ++i) {
                const InputPack &ip = ipack[i];
                if (!ip.width) continue;
                const uint8_t *src = ch->batches[i]->bc.data();
                uint8_t *dst = base + ip.col;
                if (n_skip == 0) {
                    if (ip.width == 8) {
                        for (size_t j = 0; j < ch->n; ++j) std::memcpy(dst + j * stride, src + j * 8, 8);
                    } else {
                        for (size_t j = 0; j < ch->n; ++j) std::memcpy(dst + j * stride, src + j * ip.width, ip.width);
                    }
                } else {
                    size_t rr = 0;
                    for (size_t j = 0; j < ch->n; ++j) {
                        if (ch->skip[j]) continue;
                        std::memcpy(dst + rr * stride, src + j * ip.width, ip.width);
                        ++rr;
                    }
                }
            }
            row = ch->n - n_skip;
            if (n_skip)
                for (size_t j = 0; j < ch->n; ++j) if (!ch->skip[j]) p.rows.push_back((uint32_t)j);
        } else {
            for (size_t j = 0; j < ch->n; ++j) {
                if (ch->skip[j]) continue;
                uint8_t *dst = sb[slot].obs + row * stride;
                size_t len = 0;
                for (const SegRef &r : plan.by_type[1]) {
                    const RecBatch &b = *ch->batches[r.input];
                    size_t lo, hi;
                    segment_span(plan.rs[r.input].segments[r.seg], b.recs[j].seq_len, &lo, &hi);
                    std::memcpy(dst + len, b.seq(j) + lo, hi - lo);
                    len += hi - lo;
                }
                std::memset(dst + len, 0, stride - len);
                sb[slot].lens[row] = (uint32_t)len;
                p.rows.push_back((uint32_t)j);
                ++row;
            }
            p.identity = false;
        }
        skipped += n_skip;
        p.n_rows = row;
        total_templates += row;
        if (row > 0) {
            // read structures whose B segments do not add up to the expected barcode length: the length rules
            // decide (shorter -> unmatched, longer -> the reference's panic), so the lengths must travel
            const bool need_lens = variable_barcode || fixed_barcode_len != L;
            if (!variable_barcode && need_lens) std::fill(sb[slot].lens, sb[slot].lens + row, (uint32_t)fixed_barcode_len);
            if (fqtk_matcher_enqueue(matchers[slot % G], slot / (int)G, sb[slot].obs, (uint32_t)stride, need_lens ? sb[slot].lens : nullptr, row,
                                     sb[slot].out) != FQTK_OK)
                die(fqtk_last_error());
            p.slot = slot;
        }
        while (total_templates >= next_log) { info("demultiplexed %llu records", (unsigned long long)next_log); next_log += 1000000; }
    }
    for (int d = 0; d < kSlots; ++d) finish(pending[(k + d) % kSlots]);
    for (auto &t : readers) t.join();
    info("Finished reading input FASTQs.");
    for (size_t w = 0; w < n_workers; ++w) wq[w]->push(nullptr);
    for (auto &t : workers) t.join();
    for (size_t c = 0; c < n_comp; ++c) jobs[c]->push(CompressJob{});
    for (auto &t : compressors) t.join();
    for (OutFile &of : outs) {   // every block is on disk: terminate the BGZF streams
        if (!of.ready.empty() || of.next_write != of.next_submit) die("internal error: unwritten blocks in " + of.path);
        if (std::fwrite(kBgzfEof, 1, sizeof kBgzfEof, of.f) != sizeof kBgzfEof) die("write failed: " + of.path);
        if (std::fclose(of.f) != 0) die("close failed: " + of.path);
        of.f = nullptr;
    }
    info("Output FASTQ writing complete.");
    if (g_timing)
        info("thread-seconds: routers(%zu) wait %.2f format %.2f submit %.2f | compressors(%zu) wait %.2f deflate %.2f write %.2f",
             n_workers, g_times.router_wait / 1e9, g_times.router_format / 1e9, g_times.router_submit / 1e9, n_comp,
             g_times.comp_wait / 1e9, g_times.comp_deflate / 1e9, g_times.comp_write / 1e9);
    {
        uint64_t fw = 0;
        for (auto &q : jobs) fw += q->full_waits;
        if (g_timing)
            info("main thread: waiting for readers %.2f s, for the GPU %.2f s, handing chunks to routers %.2f s | readers: parse %.2f s, push %.2f s",
                 g_times.main_wait / 1e9, g_times.main_gpu_wait / 1e9, g_times.main_handoff / 1e9, g_times.reader_parse / 1e9, g_times.reader_push / 1e9);
        if (g_timing)
            info("submit: %llu blocks, cut %.2f s (of it waiting for a free slab %.2f s), push %.2f s, %llu pushes found their queue full",
                 (unsigned long long)g_times.submit_calls.load(), g_times.submit_cut / 1e9, g_times.submit_slab / 1e9, g_times.submit_push / 1e9,
                 (unsigned long long)fw);
    }
    if (skipped == 0) info("No records were skipped.");
    else info("%llu records were skipped due to Too few bases", (unsigned long long)skipped);
    report_footprint(n_outs);

    // ---- metrics (demux.rs:994-998): counts come from the device-side per-sample histogram --------
    std::vector<uint64_t> counts(S + 1, 0);
    bool distinct = true;
    for (size_t a = 0; a < G; ++a)
        for (size_t b = 0; b < a; ++b) distinct = distinct && opt.devices[a] != opt.devices[b];
    bool reduced = false;
    if (G > 1 && distinct) {
        // the one collective of the path: per-device count vectors summed with RCCL over xGMI.  A box
        // without librccl, or a failing collective, must not cost a finished run its outputs: the accumulators are
        // only reset on success, so the host-side sum below still sees them.
        if (fqtk_matchers_allreduce_counts(matchers.data(), (int)G, 0, counts.data()) == FQTK_OK) {
            reduced = true;
            info("Per-sample counts all-reduced over %zu devices (RCCL).", G);
        } else {
            info("RCCL all-reduce of the per-sample counts unavailable (%s): summing the per-device counts on the host.", fqtk_last_error());
            std::fill(counts.begin(), counts.end(), 0);
        }
    }
    if (!reduced)
        for (fqtk_matcher *mt : matchers)   // fqtk_matcher_counts ADDS into `counts`
            if (fqtk_matcher_counts(mt, counts.data()) != FQTK_OK) die(fqtk_last_error());
    uint64_t sum = 0;
    for (uint64_t c : counts) sum += c;
    if (sum != total_templates) die("internal error: device counts do not add up to the number of templates");
    std::vector<DemuxMetric> rows(S);
    for (size_t s = 0; s < S; ++s) {
        rows[s].sample_id = samples[s].sample_id;
        rows[s].barcode = samples[s].barcode;
        rows[s].templates = counts[s];
    }
    DemuxMetric unmatched;
    unmatched.sample_id = opt.unmatched_prefix;
    unmatched.barcode = ".";
    unmatched.templates = counts[S];
    update_metrics(rows, unmatched);
    rows.push_back(unmatched);
    std::string err;
    if (!write_metrics_tsv(opt.output + "/demux-metrics.txt", rows, &err)) die(err);
    // Everything is on disk and closed.  The process ends here: tearing the HIP runtime down through the
    // matcher's destructor and the atexit handlers costs ~0.25 s and frees nothing the OS does not free.
    end_process();
}
