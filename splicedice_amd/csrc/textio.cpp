// Host-side table I/O behind the same C ABI (text I/O on both sides of the
// hot path).  Multithreaded formatter / parser for the reference's inter-stage tables
//     cluster<TAB>s0<TAB>s1...            (header line)
//     name<TAB>v<TAB>v...                 (one line per junction)
// byte-compatible with the reference's writers:
//   f'{x:.3f}' on float32 / float64   (SPLICEDICE.py:353, counts_to_ps.py:69)
//   f'{x:.0f}' on float32 counts      (SPLICEDICE.py:340)
//   str(numpy.float64)                (pairwise_fisher.py:200: shortest round-trip repr,
//                                      positional for 1e-4 <= |x| < 1e16, else d.ddde+XX)
// and with its readers (numpy string -> float64 -> dtype; compareSampleSets.py:202,
// pairwise_fisher.py:60, counts_to_ps.py:50).  No device code here.
#include <atomic>
#include <cerrno>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <string>
#include <memory>
#include <mutex>
#include <chrono>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include "sdice.h"

void sdice_set_error(const char* fmt, ...);

// seconds spent formatting / writing and bytes written by the table writers since the last reset (a measurement aid of
// tools/bench_cli.py; the calling thread's totals)
static thread_local double g_io_stats[3] = {0.0, 0.0, 0.0};
extern "C" int sdice_textio_stats(double* out3, int reset) {
    if (out3) { out3[0] = g_io_stats[0]; out3[1] = g_io_stats[1]; out3[2] = g_io_stats[2]; }
    if (reset) g_io_stats[0] = g_io_stats[1] = g_io_stats[2] = 0.0;
    return SDICE_OK;
}

namespace {

inline void put_fixed3(std::string& out, double x) {
    // '%.3f' of x: exact fast path for 0 <= x < 1000 via k = x*1000 when that product is exact
    if (std::isnan(x)) { out += "nan"; return; }
    if (x >= 0.0 && x < 1000.0) {
        const float xf = (float)x;
        if ((double)xf == x) {                          // float32 input: x*1000 is exact in double
            const double k = std::nearbyint(x * 1000.0);   // default rounding mode: half to even, as printf
            const uint32_t v = (uint32_t)k;
            char buf[16];
            int n = 0;
            uint32_t ip = v / 1000, fp = v % 1000;
            char tmp[8];
            int t = 0;
            do { tmp[t++] = (char)('0' + ip % 10); ip /= 10; } while (ip);
            if (std::signbit(x)) buf[n++] = '-';
            while (t) buf[n++] = tmp[--t];
            buf[n++] = '.';
            buf[n++] = (char)('0' + fp / 100);
            buf[n++] = (char)('0' + (fp / 10) % 10);
            buf[n++] = (char)('0' + fp % 10);
            out.append(buf, n);
            return;
        }
    }
    char buf[512];
    const int n = snprintf(buf, sizeof(buf), "%.3f", x);
    out.append(buf, n);
}

inline void put_fixed0(std::string& out, double x) {
    if (x >= 0.0 && x < 4294967296.0 && x == std::floor(x)) {
        char buf[16];
        auto r = std::to_chars(buf, buf + sizeof(buf), (uint32_t)x);
        out.append(buf, r.ptr - buf);
        return;
    }
    if (std::isnan(x)) { out += "nan"; return; }
    char buf[512];
    const int n = snprintf(buf, sizeof(buf), "%.0f", x);
    out.append(buf, n);
}

template <typename T>
inline void put_repr(std::string& out, T x) {
    // numpy str(float32/float64) == Python repr rule on the shortest round-trip digits
    if (std::isnan(x)) { out += "nan"; return; }
    if (std::isinf(x)) { out += x < 0 ? "-inf" : "inf"; return; }
    if (x == 0) { out += std::signbit(x) ? "-0.0" : "0.0"; return; }
    char sci[64];
    auto r = std::to_chars(sci, sci + sizeof(sci), x, std::chars_format::scientific);   // d[.ddd]e[+-]XX, shortest
    *r.ptr = 0;
    const char* e = strchr(sci, 'e');
    const int exp10 = atoi(e + 1);
    // digits without sign / point
    char digits[40];
    int nd = 0;
    const char* p = sci;
    const bool neg = *p == '-';
    if (neg) ++p;
    for (; p < e; ++p)
        if (*p != '.') digits[nd++] = *p;
    if (neg) out += '-';
    if (exp10 >= -4 && exp10 < 16) {
        if (exp10 >= 0) {
            for (int i = 0; i <= exp10; ++i) out += i < nd ? digits[i] : '0';
            out += '.';
            if (nd > exp10 + 1) out.append(digits + exp10 + 1, nd - exp10 - 1);
            else out += '0';
        } else {
            out += "0.";
            for (int i = 0; i < -exp10 - 1; ++i) out += '0';
            out.append(digits, nd);
        }
    } else {
        out += digits[0];
        if (nd > 1) { out += '.'; out.append(digits + 1, nd - 1); }
        out += 'e';
        out += exp10 < 0 ? '-' : '+';
        const int a = exp10 < 0 ? -exp10 : exp10;
        if (a < 10) out += '0';
        char eb[8];
        auto r2 = std::to_chars(eb, eb + sizeof(eb), a);
        out.append(eb, r2.ptr - eb);
    }
}

// numpy str(float32 / float64) straight into a character buffer (at most 26 characters): the rule of put_repr on raw
// pointers -- the per-character std::string appends of put_repr were two thirds of the time of the `pairwise` writer
template <typename T>
inline char* repr_to(char* o, T x) {
    if (std::isnan(x)) { memcpy(o, "nan", 3); return o + 3; }
    if (std::isinf(x)) { if (x < 0) *o++ = '-'; memcpy(o, "inf", 3); return o + 3; }
    if (x == 0) { if (std::signbit(x)) *o++ = '-'; memcpy(o, "0.0", 3); return o + 3; }
    char sci[48];
    auto r = std::to_chars(sci, sci + sizeof(sci), x, std::chars_format::scientific);   // [-]d[.ddd]e[+-]XX[X], shortest
    const char* p = sci;
    if (*p == '-') *o++ = *p++;
    const char* e = r.ptr - 1;
    while (*e != 'e') --e;
    int a = 0;
    for (const char* q = e + 2; q < r.ptr; ++q) a = a * 10 + (*q - '0');
    const int exp10 = e[1] == '-' ? -a : a;
    const char d0 = p[0];
    const char* frac = p + 2;
    const int nf = p[1] == '.' ? (int)(e - frac) : 0;            // digits behind the point
    if (exp10 >= -4 && exp10 < 16) {
        if (exp10 >= 0) {
            *o++ = d0;
            const int take = nf < exp10 ? nf : exp10;
            memcpy(o, frac, (size_t)take); o += take;
            for (int i = take; i < exp10; ++i) *o++ = '0';
            *o++ = '.';
            if (nf > exp10) { memcpy(o, frac + exp10, (size_t)(nf - exp10)); o += nf - exp10; }
            else *o++ = '0';
        } else {
            *o++ = '0'; *o++ = '.';
            for (int i = 0; i < -exp10 - 1; ++i) *o++ = '0';
            *o++ = d0;
            memcpy(o, frac, (size_t)nf); o += nf;
        }
    } else {
        *o++ = d0;
        if (nf) { *o++ = '.'; memcpy(o, frac, (size_t)nf); o += nf; }
        *o++ = 'e';
        *o++ = exp10 < 0 ? '-' : '+';
        if (a < 10) *o++ = '0';
        auto r2 = std::to_chars(o, o + 8, a);
        o = r2.ptr;
    }
    return o;
}

// worker threads by default: the hardware's, capped by the cgroup's CPU quota (a one-GPU box of the pool reports 256 hardware
// threads and grants 16 cores: 64 formatter threads on 16 cores ran slower than 16)
}  // namespace
int sd_default_threads() {
    static const int cached = [] {
        int n = (int)std::thread::hardware_concurrency();
        if (n < 1) n = 1;
        FILE* fh = fopen("/sys/fs/cgroup/cpu.max", "r");
        if (fh) {
            char q[32] = {0};
            long long period = 0;
            if (fscanf(fh, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
                const long long quota = atoll(q);
                if (quota > 0) n = std::min<long long>(n, (quota + period - 1) / period);
            }
            fclose(fh);
        }
        return n < 1 ? 1 : n;
    }();
    return cached;
}
namespace {
inline int default_threads() { return sd_default_threads(); }

// An exception inside a worker (std::bad_alloc from a growing row buffer) or from a thread that cannot be started
// must not reach std::terminate: workers catch into `err`, every started thread is joined, the first exception is
// rethrown on the caller's thread (the extern "C" function-try-blocks map it to SDICE_ERR_*).
template <typename F>
void parallel_rows(int64_t n, int threads, F&& fn) {
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    if (n < threads) threads = (int)std::max<int64_t>(n, 1);      // (the CALLER decides whether the work is worth threads)
    if (threads == 1) { fn(0, 0, n); return; }
    std::vector<std::thread> pool;
    std::exception_ptr err;
    std::mutex err_mu;
    try {
        for (int t = 0; t < threads; ++t) {
            const int64_t a = n * t / threads, b = n * (t + 1) / threads;
            pool.emplace_back([=, &fn, &err, &err_mu] {
                try { fn(t, a, b); }
                catch (...) { std::lock_guard<std::mutex> g(err_mu); if (!err) err = std::current_exception(); }
            });
        }
    } catch (...) {
        std::lock_guard<std::mutex> g(err_mu);
        if (!err) err = std::current_exception();
    }
    for (auto& th : pool) th.join();
    if (err) std::rethrow_exception(err);
}

}  // namespace

// mode: 0 = '%.3f', 1 = '%.0f', 2 = numpy str() shortest repr; + 0x100: append to an existing file (a table
//       streamed in row slabs -- `pairwise` at config-4 size writes 32 GB of p-values this way)
// dtype: 0 = float32, 1 = float64, 2 = int32 (mode 1 only)
namespace {
struct FdCloser {
    int fd;
    ~FdCloser() { if (fd >= 0) close(fd); }
};
// all of [p, p + len) at file offset `at` (pwrite may stop short; EINTR is retried)
inline bool write_all_at(int fd, const char* p, size_t len, off_t at) {
    while (len > 0) {
        const ssize_t w = pwrite(fd, p, len, at);
        if (w < 0) { if (errno == EINTR) continue; return false; }
        if (w == 0) return false;
        p += w; len -= (size_t)w; at += w;
    }
    return true;
}
// text buffers of the table writer's threads, kept between calls (sdice_textio_trim frees them)
struct WBufPool {
    std::mutex mu;
    std::vector<std::unique_ptr<char[]>> buf;
    std::vector<size_t> cap;
};
WBufPool g_wpool;
}  // namespace

// frees the writer's text buffers (they are kept between calls: see sdice_write_table)
extern "C" int sdice_textio_trim(void) {
    std::lock_guard<std::mutex> g(g_wpool.mu);
    g_wpool.buf.clear();
    g_wpool.cap.clear();
    g_wpool.buf.shrink_to_fit();
    return SDICE_OK;
}

extern "C" int sdice_write_table(const char* path, const char* header, int64_t n, int32_t s, const char* names,
                                 const int64_t* name_off, const void* data, int dtype, int mode, int threads) try {
    if (!path || !header || n < 0 || s < 0 || (n > 0 && (!names || !name_off || (!data && s > 0)))) {
        sdice_set_error("sdice_write_table: bad arguments");
        return SDICE_ERR_ARG;
    }
    const bool append = (mode & 0x100) != 0;
    mode &= 0xff;
    if (dtype < 0 || dtype > 2 || mode < 0 || mode > 2 || (dtype == 2 && mode != 1)) {
        sdice_set_error("sdice_write_table: unsupported dtype/mode combination");
        return SDICE_ERR_ARG;
    }
    // The text leaves through a file DESCRIPTOR: every formatting thread writes its own piece at its own offset
    // (pwrite: the copy into the page cache, ~10 GB per `pairwise` table, was one thread's work behind fwrite).
    FdCloser fd{open(path, O_WRONLY | O_CREAT | (append ? 0 : O_TRUNC), 0666)};
    if (fd.fd < 0) {
        sdice_set_error("sdice_write_table: cannot open %s", path);
        return SDICE_ERR_ARG;
    }
    off_t file_pos = append ? lseek(fd.fd, 0, SEEK_END) : 0;
    if (file_pos < 0 || !write_all_at(fd.fd, header, strlen(header), file_pos)) {
        sdice_set_error("sdice_write_table: short write to %s", path);
        return SDICE_ERR_ARG;
    }
    file_pos += (off_t)strlen(header);
    // rows are formatted in blocks so that memory stays bounded and the writes stay ordered
    const int64_t block = 1 << 16;
    int nthreads = threads > 0 ? threads : default_threads();
    // the threads' text buffers outlive the call (a `pairwise` table comes in ~60 slabs of ~1 GB of text each: fresh
    // buffers per slab are a quarter of a million page faults per slab); a second writer at the same time brings its own
    std::unique_lock<std::mutex> pool_lock(g_wpool.mu, std::try_to_lock);
    const bool pooled = pool_lock.owns_lock();
    std::vector<std::unique_ptr<char[]>> own_bufs;
    std::vector<size_t> own_caps;
    std::vector<std::unique_ptr<char[]>>& bufs = pooled ? g_wpool.buf : own_bufs;
    std::vector<size_t>& caps = pooled ? g_wpool.cap : own_caps;
    for (int64_t r0 = 0; r0 < n; r0 += block) {
        const int64_t nb = std::min(block, n - r0);
        // threads by CELLS, not rows: a `pairwise` slab is ~1 700 rows of 19 900 columns (deciding by rows left that
        // writer on ONE thread: 1.5e7 values/s through the CLI in round 3)
        int used = nb * (int64_t)std::max(s, 1) < 65536 ? 1 : std::min(nthreads, 64);
        if (used < 1) used = 1;
        if ((int64_t)used > nb) used = (int)nb;
        if (bufs.size() < (size_t)used) { bufs.resize((size_t)used); caps.resize((size_t)used, 0); }
        // The block goes out in CHUNKS dealt round-robin to the threads: a thread formats chunk c, waits until the chunk
        // before it has published where it ends (off[c]), publishes its own end, and writes its text there itself.
        // Buffered writes to one file take turns inside the kernel (the inode lock: 16 threads writing at once moved
        // the same 9.4 GB/s as one), so the gain is not parallel writing but that the other threads keep FORMATTING
        // while one is in write(): per `pairwise` table 1.5 s of formatting and 1.05 s of writing, back to back before.
        const int64_t n_chunks = std::min<int64_t>(nb, (int64_t)used * 8);
        std::vector<std::atomic<int64_t>> off((size_t)n_chunks + 1);
        for (auto& o : off) o.store(-1, std::memory_order_relaxed);
        off[0].store((int64_t)file_pos, std::memory_order_release);
        std::atomic<bool> failed{false};
        std::vector<double> fmt_s((size_t)used, 0.0), wr_s((size_t)used, 0.0);
        parallel_rows(used, used, [&](int, int64_t t_lo, int64_t t_hi) {
          for (int64_t t = t_lo; t < t_hi; ++t) {
            try {
                std::string tmp;
                for (int64_t ch = t; ch < n_chunks && !failed.load(std::memory_order_relaxed); ch += used) {
                    const auto t0 = std::chrono::steady_clock::now();
                    const int64_t a = nb * ch / n_chunks, b = nb * (ch + 1) / n_chunks;
                    // worst case per cell: tab + 26 characters (numpy repr of a double, '%.3f' / '%.0f' of a count or PS
                    // value below 1e21; anything longer takes the slow path through a temporary) -- written with a bare pointer
                    size_t cap = (size_t)(name_off[r0 + b] - name_off[r0 + a]) + (size_t)(b - a) * ((size_t)s * 27 + 1) + 64;
                    std::unique_ptr<char[]>& raw = bufs[(size_t)t];          // (slot t is this thread's for the block)
                    if (caps[(size_t)t] < cap) { raw.reset(); raw.reset(new char[cap]); caps[(size_t)t] = cap; }
                    cap = caps[(size_t)t];
                    char* o = raw.get();
                    for (int64_t r = r0 + a; r < r0 + b; ++r) {
                        const size_t nl = (size_t)(name_off[r + 1] - name_off[r]);
                        memcpy(o, names + name_off[r], nl); o += nl;
                        for (int32_t c = 0; c < s; ++c) {
                            *o++ = '\t';
                            const size_t i = (size_t)r * (size_t)s + (size_t)c;
                            if (dtype == 2) {
                                auto rr = std::to_chars(o, o + 16, ((const int32_t*)data)[i]);
                                o = rr.ptr;
                            } else if (mode == 2) {
                                o = dtype == 0 ? repr_to<float>(o, ((const float*)data)[i]) : repr_to<double>(o, ((const double*)data)[i]);
                            } else {
                                const double v = dtype == 0 ? (double)((const float*)data)[i] : ((const double*)data)[i];
                                tmp.clear();
                                if (mode == 0) put_fixed3(tmp, v); else put_fixed0(tmp, v);
                                if (tmp.size() > 26) {                      // (a count beyond 1e21: grow the buffer)
                                    const size_t used_b = (size_t)(o - raw.get());
                                    cap += tmp.size() + 64;
                                    std::unique_ptr<char[]> bigger(new char[cap + tmp.size()]);
                                    memcpy(bigger.get(), raw.get(), used_b);
                                    raw.swap(bigger);
                                    caps[(size_t)t] = cap + tmp.size();
                                    o = raw.get() + used_b;
                                }
                                memcpy(o, tmp.data(), tmp.size()); o += tmp.size();
                            }
                        }
                        *o++ = '\n';
                    }
                    const int64_t len = (int64_t)(o - raw.get());
                    const auto t1 = std::chrono::steady_clock::now();
                    int64_t at;
                    while ((at = off[(size_t)ch].load(std::memory_order_acquire)) < 0) {
                        if (failed.load(std::memory_order_relaxed)) break;
                        std::this_thread::yield();
                    }
                    if (at < 0) break;
                    off[(size_t)ch + 1].store(at + len, std::memory_order_release);     // (before the write: the next chunk may go)
                    if (len && !write_all_at(fd.fd, raw.get(), (size_t)len, (off_t)at)) failed.store(true);
                    const auto t2 = std::chrono::steady_clock::now();
                    fmt_s[(size_t)t] += std::chrono::duration<double>(t1 - t0).count();
                    wr_s[(size_t)t] += std::chrono::duration<double>(t2 - t1).count();
                }
            } catch (...) {
                failed.store(true);            // (the threads that wait for this one's offsets give up)
                throw;
            }
          }
        });
        const int64_t end_pos = off[(size_t)n_chunks].load(std::memory_order_acquire);
        if (failed.load() || end_pos < 0) {
            sdice_set_error("sdice_write_table: short write to %s", path);
            return SDICE_ERR_ARG;
        }
        g_io_stats[2] += (double)(end_pos - (int64_t)file_pos);
        file_pos = (off_t)end_pos;
        // (formatting and writing overlap now: the split reported is the threads' mean time in either)
        double fs = 0.0, ws = 0.0;
        for (int t = 0; t < used; ++t) { fs += fmt_s[(size_t)t]; ws += wr_s[(size_t)t]; }
        g_io_stats[0] += fs / used;
        g_io_stats[1] += ws / used;
    }
    const auto t_cl = std::chrono::steady_clock::now();
    const int fdv = fd.fd;
    fd.fd = -1;
    if (close(fdv) != 0) {
        sdice_set_error("sdice_write_table: close failed for %s", path);
        return SDICE_ERR_ARG;
    }
    g_io_stats[1] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_cl).count();
    return SDICE_OK;
} catch (const std::exception& e) {
    sdice_set_error("sdice_write_table: %s", e.what());
    return SDICE_ERR_NOMEM;
} catch (...) {
    sdice_set_error("sdice_write_table: unknown exception");
    return SDICE_ERR_STATE;
}

// Column-major variant: every column has its own array, dtype and mode (the compare_sample_sets
// output mixes float32 and float64 numpy-repr columns, compareSampleSets.py:252-270).
static int write_columns_impl(const char* path, const char* header, int64_t n, const char* names,
                              const int64_t* name_off, int32_t ncols, const void* const* cols,
                              const int32_t* dtypes, const int32_t* modes, const char* sfx, const int64_t* sfx_off,
                              int threads) {
    if (!path || !header || n < 0 || ncols < 0 || (n > 0 && (!names || !name_off)) || (ncols > 0 && (!cols || !dtypes || !modes)) ||
        (sfx_off && !sfx && n > 0 && sfx_off[n] > 0)) {
        sdice_set_error("sdice_write_columns: bad arguments");
        return SDICE_ERR_ARG;
    }
    for (int32_t c = 0; c < ncols; ++c)
        if (!cols[c] || dtypes[c] < 0 || dtypes[c] > 2 || modes[c] < 0 || modes[c] > 2 || (dtypes[c] == 2 && modes[c] != 1)) {
            sdice_set_error("sdice_write_columns: unsupported dtype/mode in column %d", (int)c);
            return SDICE_ERR_ARG;
        }
    FILE* fh = fopen(path, "wb");
    if (!fh) {
        sdice_set_error("sdice_write_columns: cannot open %s", path);
        return SDICE_ERR_ARG;
    }
    fwrite(header, 1, strlen(header), fh);
    const int64_t block = 1 << 16;
    int nthreads = threads > 0 ? threads : default_threads();
    std::vector<std::string> bufs;
    for (int64_t r0 = 0; r0 < n; r0 += block) {
        const int64_t nb = std::min(block, n - r0);
        int used = nb < 4096 ? 1 : std::min(nthreads, 64);
        if (used < 1) used = 1;
        bufs.assign(used, std::string());
        parallel_rows(nb, used, [&](int t, int64_t a, int64_t b) {
            std::string& out = bufs[t];
            out.reserve((size_t)(b - a) * ((size_t)ncols * 20 + 32));
            for (int64_t r = r0 + a; r < r0 + b; ++r) {
                out.append(names + name_off[r], (size_t)(name_off[r + 1] - name_off[r]));
                for (int32_t c = 0; c < ncols; ++c) {
                    out += '\t';
                    if (dtypes[c] == 2) {
                        char b2[16];
                        auto rr = std::to_chars(b2, b2 + sizeof(b2), ((const int32_t*)cols[c])[r]);
                        out.append(b2, rr.ptr - b2);
                    } else if (dtypes[c] == 0) {
                        const float v = ((const float*)cols[c])[r];
                        if (modes[c] == 0) put_fixed3(out, (double)v);
                        else if (modes[c] == 1) put_fixed0(out, (double)v);
                        else put_repr<float>(out, v);
                    } else {
                        const double v = ((const double*)cols[c])[r];
                        if (modes[c] == 0) put_fixed3(out, v);
                        else if (modes[c] == 1) put_fixed0(out, v);
                        else put_repr<double>(out, v);
                    }
                }
                if (sfx_off) out.append(sfx + sfx_off[r], (size_t)(sfx_off[r + 1] - sfx_off[r]));
                out += '\n';
            }
        });
        for (auto& b : bufs)
            if (!b.empty() && fwrite(b.data(), 1, b.size(), fh) != b.size()) {
                fclose(fh);
                sdice_set_error("sdice_write_columns: short write to %s", path);
                return SDICE_ERR_ARG;
            }
    }
    if (fclose(fh) != 0) {
        sdice_set_error("sdice_write_columns: close failed for %s", path);
        return SDICE_ERR_ARG;
    }
    return SDICE_OK;
}

extern "C" int sdice_write_columns(const char* path, const char* header, int64_t n, const char* names,
                                   const int64_t* name_off, int32_t ncols, const void* const* cols,
                                   const int32_t* dtypes, const int32_t* modes, int threads) try {
    return write_columns_impl(path, header, n, names, name_off, ncols, cols, dtypes, modes, nullptr, nullptr, threads);
} catch (const std::exception& e) {
    sdice_set_error("sdice_write_columns: %s", e.what());
    return SDICE_ERR_NOMEM;
} catch (...) {
    sdice_set_error("sdice_write_columns: unknown exception");
    return SDICE_ERR_STATE;
}

// The same table with a ready-made text suffix per row (sfx[sfx_off[r] .. sfx_off[r+1]), written after the last
// numeric column, before the newline): the gene / overlapping / transcript_id columns of an annotated
// compare_sample_sets table (compareSampleSets.py:238-264).
extern "C" int sdice_write_columns_sfx(const char* path, const char* header, int64_t n, const char* names,
                                       const int64_t* name_off, int32_t ncols, const void* const* cols,
                                       const int32_t* dtypes, const int32_t* modes, const char* sfx,
                                       const int64_t* sfx_off, int threads) try {
    if (!sfx_off) { sdice_set_error("sdice_write_columns_sfx: sfx_off is NULL"); return SDICE_ERR_ARG; }
    return write_columns_impl(path, header, n, names, name_off, ncols, cols, dtypes, modes, sfx, sfx_off, threads);
} catch (const std::exception& e) {
    sdice_set_error("sdice_write_columns_sfx: %s", e.what());
    return SDICE_ERR_NOMEM;
} catch (...) {
    sdice_set_error("sdice_write_columns_sfx: unknown exception");
    return SDICE_ERR_STATE;
}

// `_allClusters.tsv` (SPLICEDICE.py:316-326): one line per junction row,
// name<TAB>name_of_neighbour_1,name_of_neighbour_2,...  (a junction without overlaps: name<TAB>).
extern "C" int sdice_write_clusters(const char* path, int64_t n, const char* names, const int64_t* name_off,
                                    const int64_t* row_ptr, const int32_t* col, int threads) try {
    if (!path || n < 0 || (n > 0 && (!names || !name_off || !row_ptr)) || (n > 0 && row_ptr[n] > 0 && !col)) {
        sdice_set_error("sdice_write_clusters: bad arguments");
        return SDICE_ERR_ARG;
    }
    FILE* fh = fopen(path, "wb");
    if (!fh) {
        sdice_set_error("sdice_write_clusters: cannot open %s", path);
        return SDICE_ERR_ARG;
    }
    const int64_t block = 1 << 16;
    int nthreads = threads > 0 ? threads : default_threads();
    std::vector<std::string> bufs;
    for (int64_t r0 = 0; r0 < n; r0 += block) {
        const int64_t nb = std::min(block, n - r0);
        int used = nb < 4096 ? 1 : std::min(nthreads, 64);
        if (used < 1) used = 1;
        bufs.assign(used, std::string());
        std::atomic<bool> bad{false};       // (set by any worker thread)
        parallel_rows(nb, used, [&](int t, int64_t a, int64_t b) {
            std::string& out = bufs[t];
            for (int64_t r = r0 + a; r < r0 + b; ++r) {
                out.append(names + name_off[r], (size_t)(name_off[r + 1] - name_off[r]));
                out += '\t';
                for (int64_t k = row_ptr[r]; k < row_ptr[r + 1]; ++k) {
                    const int64_t c = col[k];
                    if (c < 0 || c >= n) { bad = true; continue; }
                    if (k > row_ptr[r]) out += ',';
                    out.append(names + name_off[c], (size_t)(name_off[c + 1] - name_off[c]));
                }
                out += '\n';
            }
        });
        if (bad) {
            fclose(fh);
            sdice_set_error("sdice_write_clusters: col index out of range");
            return SDICE_ERR_ARG;
        }
        for (auto& b : bufs)
            if (!b.empty() && fwrite(b.data(), 1, b.size(), fh) != b.size()) {
                fclose(fh);
                sdice_set_error("sdice_write_clusters: short write to %s", path);
                return SDICE_ERR_ARG;
            }
    }
    if (fclose(fh) != 0) {
        sdice_set_error("sdice_write_clusters: close failed for %s", path);
        return SDICE_ERR_ARG;
    }
    return SDICE_OK;
} catch (const std::exception& e) {
    sdice_set_error("sdice_write_clusters: %s", e.what());
    return SDICE_ERR_NOMEM;
} catch (...) {
    sdice_set_error("sdice_write_clusters: unknown exception");
    return SDICE_ERR_STATE;
}

// `_junctions.bed` (SPLICEDICE.py:316-321): one line per junction row,
// chrom<TAB>left<TAB>right<TAB>chrom:left-right:strand<TAB>0<TAB>strand
extern "C" int sdice_write_junction_bed(const char* path, int64_t n, const char* chrom_names, const int64_t* chrom_off,
                                        int32_t n_chrom, const int32_t* chrom, const int32_t* left, const int32_t* right,
                                        const char* strand, int threads) try {
    if (!path || n < 0 || n_chrom < 0 || (n > 0 && (!chrom_names || !chrom_off || !chrom || !left || !right || !strand))) {
        sdice_set_error("sdice_write_junction_bed: bad arguments");
        return SDICE_ERR_ARG;
    }
    FILE* fh = fopen(path, "wb");
    if (!fh) {
        sdice_set_error("sdice_write_junction_bed: cannot open %s", path);
        return SDICE_ERR_ARG;
    }
    const int64_t block = 1 << 18;
    int nthreads = threads > 0 ? threads : default_threads();
    std::vector<std::string> bufs;
    for (int64_t r0 = 0; r0 < n; r0 += block) {
        const int64_t nb = std::min(block, n - r0);
        int used = nb < 4096 ? 1 : std::min(nthreads, 64);
        if (used < 1) used = 1;
        bufs.assign(used, std::string());
        std::atomic<bool> bad{false};
        parallel_rows(nb, used, [&](int t, int64_t a, int64_t b) {
            std::string& out = bufs[t];
            char num[16];
            for (int64_t r = r0 + a; r < r0 + b; ++r) {
                const int32_t c = chrom[r];
                if (c < 0 || c >= n_chrom) { bad = true; continue; }
                const char* cn = chrom_names + chrom_off[c];
                const size_t cl = (size_t)(chrom_off[c + 1] - chrom_off[c]);
                const int ll = snprintf(num, sizeof num, "%d", left[r]);
                const std::string ls(num, (size_t)ll);
                const int rl = snprintf(num, sizeof num, "%d", right[r]);
                const std::string rs(num, (size_t)rl);
                out.append(cn, cl); out += '\t'; out += ls; out += '\t'; out += rs; out += '\t';
                out.append(cn, cl); out += ':'; out += ls; out += '-'; out += rs; out += ':'; out += strand[r];
                out += "\t0\t"; out += strand[r]; out += '\n';
            }
        });
        if (bad) {
            fclose(fh);
            sdice_set_error("sdice_write_junction_bed: chromosome index out of range");
            return SDICE_ERR_ARG;
        }
        for (auto& b : bufs)
            if (!b.empty() && fwrite(b.data(), 1, b.size(), fh) != b.size()) {
                fclose(fh);
                sdice_set_error("sdice_write_junction_bed: short write to %s", path);
                return SDICE_ERR_ARG;
            }
    }
    if (fclose(fh) != 0) {
        sdice_set_error("sdice_write_junction_bed: close failed for %s", path);
        return SDICE_ERR_ARG;
    }
    return SDICE_OK;
} catch (const std::exception& e) {
    sdice_set_error("sdice_write_junction_bed: %s", e.what());
    return SDICE_ERR_NOMEM;
} catch (...) {
    sdice_set_error("sdice_write_junction_bed: unknown exception");
    return SDICE_ERR_STATE;
}

// Row names 'chrom:left-right:strand' (SPLICEDICE.py:312-314) of n junction rows as one byte string + offsets, for the
// table writers above (which take names in exactly this form): out (room for out_cap bytes) and off[n + 1].
// strand: the symbol per row.  *need = bytes the names take (also when out_cap is too small: SDICE_ERR_ARG then).
extern "C" int sdice_junction_names(int64_t n, const char* chrom_names, const int64_t* chrom_off, int32_t n_chrom,
                                    const int32_t* chrom, const int32_t* left, const int32_t* right, const char* strand,
                                    char* out, int64_t out_cap, int64_t* off, int64_t* need) try {
    if (n < 0 || n_chrom < 0 || !off || (n > 0 && (!chrom_names || !chrom_off || !chrom || !left || !right || !strand))) {
        sdice_set_error("sdice_junction_names: bad arguments");
        return SDICE_ERR_ARG;
    }
    int64_t pos = 0;
    bool fits = out != nullptr;
    char num[16];
    for (int64_t r = 0; r < n; ++r) {
        const int32_t c = chrom[r];
        if (c < 0 || c >= n_chrom) { sdice_set_error("sdice_junction_names: chromosome index out of range"); return SDICE_ERR_ARG; }
        const int64_t cl = chrom_off[c + 1] - chrom_off[c];
        off[r] = pos;
        const auto l_end = std::to_chars(num, num + sizeof num, left[r]).ptr;
        const int64_t ll = l_end - num;
        char num2[16];
        const int64_t rl = std::to_chars(num2, num2 + sizeof num2, right[r]).ptr - num2;
        const int64_t len = cl + 1 + ll + 1 + rl + 2;
        if (fits && pos + len <= out_cap) {
            char* o = out + pos;
            memcpy(o, chrom_names + chrom_off[c], (size_t)cl); o += cl;
            *o++ = ':'; memcpy(o, num, (size_t)ll); o += ll;
            *o++ = '-'; memcpy(o, num2, (size_t)rl); o += rl;
            *o++ = ':'; *o++ = strand[r];
        } else {
            fits = false;
        }
        pos += len;
    }
    off[n] = pos;
    if (need) *need = pos;
    if (!fits && n > 0) {
        sdice_set_error("sdice_junction_names: the names take %lld bytes, room for %lld", (long long)pos, (long long)out_cap);
        return SDICE_ERR_ARG;
    }
    return SDICE_OK;
} catch (const std::exception& e) {
    sdice_set_error("sdice_junction_names: %s", e.what());
    return SDICE_ERR_NOMEM;
} catch (...) {
    sdice_set_error("sdice_junction_names: unknown exception");
    return SDICE_ERR_STATE;
}

// ------------------------------------------------------------------------------------------ reader
struct sdice_table {
    int fd = -1;
    const char* base = nullptr;
    size_t size = 0;
    size_t body = 0;                  // offset of the first data line
    std::vector<size_t> line_start;   // n + 1 entries (last = end of data)
    int64_t n = 0;
    int32_t s = 0;
    int64_t names_bytes = 0;
    int rstrip_mode = 0;
};

extern "C" int sdice_table_close(sdice_table* t) try {
    if (!t) return SDICE_OK;
    if (t->base && t->size) munmap((void*)t->base, t->size);
    if (t->fd >= 0) close(t->fd);
    delete t;
    return SDICE_OK;
} catch (const std::exception& e) {
    sdice_set_error("sdice_table_close: %s", e.what());
    return SDICE_ERR_NOMEM;
} catch (...) {
    sdice_set_error("sdice_table_close: unknown exception");
    return SDICE_ERR_STATE;
}

// Opens a table and indexes its lines.  n = data lines, s = tab-separated value columns of the
// header (= header fields - 1), names_bytes = total length of the first field of every line.
extern "C" int sdice_table_open(const char* path, sdice_table** out, int64_t* n, int32_t* s, int64_t* names_bytes,
                                int64_t* header_bytes) try {
    if (!path || !out) { sdice_set_error("sdice_table_open: bad arguments"); return SDICE_ERR_ARG; }
    *out = nullptr;
    sdice_table* t = new sdice_table();
    t->fd = open(path, O_RDONLY);
    struct stat st;
    if (t->fd < 0 || fstat(t->fd, &st) != 0) {
        sdice_set_error("sdice_table_open: cannot open %s", path);
        sdice_table_close(t);
        return SDICE_ERR_ARG;
    }
    t->size = (size_t)st.st_size;
    if (t->size) {
        void* m = mmap(nullptr, t->size, PROT_READ, MAP_PRIVATE, t->fd, 0);
        if (m == MAP_FAILED) {
            sdice_set_error("sdice_table_open: mmap failed for %s", path);
            t->size = 0;
            sdice_table_close(t);
            return SDICE_ERR_ARG;
        }
        t->base = (const char*)m;
    }
    const char* p = t->base;
    const char* end = p + t->size;
    const char* nl = t->size ? (const char*)memchr(p, '\n', t->size) : nullptr;
    const char* hend = nl ? nl : end;
    int fields = t->size ? 1 : 0;
    for (const char* q = p; q < hend; ++q) fields += (*q == '\t');
    t->s = fields > 0 ? fields - 1 : 0;
    t->body = nl ? (size_t)(nl + 1 - p) : t->size;
    // line index (single pass; memchr is fast enough to not need threads)
    size_t pos = t->body;
    while (pos < t->size) {
        t->line_start.push_back(pos);
        const char* e = (const char*)memchr(p + pos, '\n', t->size - pos);
        pos = e ? (size_t)(e + 1 - p) : t->size;
    }
    t->line_start.push_back(t->size);
    t->n = (int64_t)t->line_start.size() - 1;
    int64_t nb = 0;
    for (int64_t i = 0; i < t->n; ++i) {
        const char* a = p + t->line_start[i];
        const char* b = p + t->line_start[i + 1];
        const char* tab = (const char*)memchr(a, '\t', (size_t)(b - a));
        const char* stop = tab ? tab : b;
        while (stop > a && (stop[-1] == '\n' || stop[-1] == '\r')) --stop;
        nb += stop - a;
    }
    t->names_bytes = nb;
    if (n) *n = t->n;
    if (s) *s = t->s;
    if (names_bytes) *names_bytes = nb;
    if (header_bytes) *header_bytes = (int64_t)t->body;
    *out = t;
    return SDICE_OK;
} catch (const std::exception& e) {
    sdice_set_error("sdice_table_open: %s", e.what());
    return SDICE_ERR_NOMEM;
} catch (...) {
    sdice_set_error("sdice_table_open: unknown exception");
    return SDICE_ERR_STATE;
}

// Fills header (header_bytes incl. newline, not NUL terminated), names blob + offsets[n+1] and
// data[n, s] (dtype 0 float32, 1 float64; numpy semantics: text -> float64 -> dtype).
// A line with a different number of fields than the header is an error, as numpy would raise.
extern "C" int sdice_table_read(sdice_table* t, char* header, char* names, int64_t* name_off, void* data, int dtype,
                                int threads) try {
    if (!t || (t->n > 0 && (!names || !name_off || (!data && t->s > 0))) || dtype < 0 || dtype > 1) {
        sdice_set_error("sdice_table_read: bad arguments");
        return SDICE_ERR_ARG;
    }
    if (header && t->body) memcpy(header, t->base, t->body);
    // name offsets first (serial prefix), then parallel parse
    int64_t off = 0;
    std::vector<const char*> name_end((size_t)t->n);
    for (int64_t i = 0; i < t->n; ++i) {
        const char* a = t->base + t->line_start[i];
        const char* b = t->base + t->line_start[i + 1];
        const char* tab = (const char*)memchr(a, '\t', (size_t)(b - a));
        const char* stop = tab ? tab : b;
        while (stop > a && (stop[-1] == '\n' || stop[-1] == '\r')) --stop;
        name_off[i] = off;
        memcpy(names + off, a, (size_t)(stop - a));
        off += stop - a;
        name_end[(size_t)i] = tab ? tab : stop;
    }
    if (t->n >= 0 && name_off) name_off[t->n] = off;
    std::vector<int64_t> bad((size_t)64, -1);
    int nthreads = threads > 0 ? threads : default_threads();
    if (nthreads > 64) nthreads = 64;
    parallel_rows(t->n, nthreads, [&](int tix, int64_t a, int64_t b) {
        for (int64_t i = a; i < b; ++i) {
            const char* p = name_end[(size_t)i];
            const char* end = t->base + t->line_start[i + 1];
            while (end > p && (end[-1] == '\n' || end[-1] == '\r' || end[-1] == ' ')) --end;   // line.rstrip()
            int32_t c = 0;
            while (p < end && c < t->s) {
                if (*p != '\t') { bad[tix] = i; break; }
                ++p;
                const char* q = (const char*)memchr(p, '\t', (size_t)(end - p));
                const char* fe = q ? q : end;
                double v = 0.0;
                const char* fs = p;
                while (fs < fe && *fs == ' ') ++fs;
                if (fs < fe && *fs == '+') ++fs;     // from_chars rejects a leading '+', float() accepts it
                auto r = std::from_chars(fs, fe, v);
                if (r.ec != std::errc() || r.ptr != fe) { bad[tix] = i; break; }
                const size_t idx = (size_t)i * (size_t)t->s + (size_t)c;
                if (dtype == 0) ((float*)data)[idx] = (float)v;
                else ((double*)data)[idx] = v;
                ++c;
                p = fe;
            }
            if (bad[tix] == i) break;
            if (c != t->s || p != end) { bad[tix] = i; break; }
        }
    });
    for (auto b : bad)
        if (b >= 0) {
            sdice_set_error("sdice_table_read: line %lld is not %d numeric tab-separated fields", (long long)(b + 2),
                            (int)t->s);
            return SDICE_ERR_ARG;
        }
    return SDICE_OK;
} catch (const std::exception& e) {
    sdice_set_error("sdice_table_read: %s", e.what());
    return SDICE_ERR_NOMEM;
} catch (...) {
    sdice_set_error("sdice_table_read: unknown exception");
    return SDICE_ERR_STATE;
}
