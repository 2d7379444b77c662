// Host-side junction-file parser behind the C ABI.
//
// Parses one sample file of `splicedice quant` -- splicedicebed / bed / leafcutter lines or a
// STAR SJ.out.tab -- with threads, and applies the reference's per-type admission filters
// (SPLICEDICE.getAllJunctions, SPLICEDICE.py:147-228) while keeping every line's key and score
// for the count pass (SPLICEDICE.getJunctionCounts, SPLICEDICE.py:257-295).  The reference reads
// every file twice with a Python loop per line; here a file is read once.
//   type 0  bed / leafcutter : chrom left right name score strand          (admit: score >= minUnique,
//                              minLength <= len <= maxLength, strand in {+,-})
//   type 1  splicedicebed    : name = e:<Lent>:<Rent>;o:<overhang>;m:..;a:<gene|?>
//                              (the filters above + overhang + both entropies apply only when a == '?')
//   type 2  SJ.out.tab       : chrom start end strand(0/1/2) motif annot unique multi overhang
//                              left = start - 1; score = unique (+ multi unless noMultimap); admit:
//                              minLength < len < maxLength, strand != 0, score >= minUnique, motif in {1,2}
// A malformed line is an error (the reference raises on it too).
#include <charconv>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include "sdice.h"

int sd_default_threads();   // textio.cpp: hardware threads capped by the cgroup's CPU quota

void sdice_set_error(const char* fmt, ...);

struct sdice_juncfile {
    int fd = -1;
    const char* base = nullptr;
    size_t size = 0;
    int type = 0;
    std::vector<size_t> line_start;     // n + 1
    int64_t n = 0;
    std::vector<std::string> chroms;    // first-appearance order
    std::vector<int32_t> chrom_id;      // per line
};

namespace {
// joins every started thread when the scope ends, also when std::thread's constructor threw half way through the pool
// (the workers themselves do not allocate and cannot throw)
struct JoinAll {
    std::vector<std::thread>& pool;
    ~JoinAll() { for (auto& t : pool) if (t.joinable()) t.join(); }
};
}  // namespace

namespace {

struct Field { const char* a; const char* b; };

inline bool split_tabs(const char* p, const char* end, Field* f, int want) {
    int k = 0;
    while (k < want) {
        const char* q = (const char*)memchr(p, '\t', (size_t)(end - p));
        f[k].a = p;
        f[k].b = q ? q : end;
        ++k;
        if (!q) break;
        p = q + 1;
    }
    return k == want;
}

inline bool to_i64(Field f, int64_t& v) {
    const char* a = f.a;
    if (a < f.b && *a == '+') ++a;
    auto r = std::from_chars(a, f.b, v);
    return r.ec == std::errc() && r.ptr == f.b;
}

// float(text) for the entropy fields.  Plain decimals of up to 15 significant digits -- "1.50", what bam_to_junc_bed
// writes -- take the exact route: the digits as an integer below 2^53 divided by a power of ten up to 10^22, both exact
// doubles, so the one IEEE division is the correctly rounded value, the same double that from_chars (and Python's float())
// returns.  Everything else (exponents, long digit strings, inf / nan, a bare '.') goes to from_chars.
inline bool to_f64(const char* a, const char* b, double& v) {
    if (a < b && *a == '+') ++a;
    static const double P10[] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};
    {
        const char* p = a;
        const bool neg = p < b && *p == '-';
        if (neg) ++p;
        uint64_t m = 0;
        int nd = 0, frac = 0;
        bool dot = false, plain = p < b;
        for (; p < b; ++p) {
            const unsigned d = (unsigned)(*p - '0');
            if (d <= 9u) { m = m * 10u + d; ++nd; frac += dot ? 1 : 0; }
            else if (*p == '.' && !dot) dot = true;
            else { plain = false; break; }
        }
        if (plain && nd >= 1 && nd <= 15) {
            const double x = (double)m / P10[frac];
            v = neg ? -x : x;
            return true;
        }
    }
    auto r = std::from_chars(a, b, v);
    return r.ec == std::errc() && r.ptr == b;
}

}  // namespace

extern "C" int sdice_junc_close(sdice_juncfile* t) try {
    if (!t) return SDICE_OK;
    if (t->base && t->size) munmap((void*)t->base, t->size);
    if (t->fd >= 0) close(t->fd);
    delete t;
    return SDICE_OK;
} catch (const std::exception& e) {
    sdice_set_error("sdice_junc_close: %s", e.what());
    return SDICE_ERR_NOMEM;
} catch (...) {
    sdice_set_error("sdice_junc_close: unknown exception");
    return SDICE_ERR_STATE;
}

extern "C" int sdice_junc_open(const char* path, int type, sdice_juncfile** out, int64_t* n_lines, int32_t* n_chroms,
                               int64_t* chrom_bytes) try {
    if (!path || !out || type < 0 || type > 2) { sdice_set_error("sdice_junc_open: bad arguments"); return SDICE_ERR_ARG; }
    *out = nullptr;
    sdice_juncfile* t = new sdice_juncfile();
    t->type = type;
    t->fd = open(path, O_RDONLY);
    struct stat st;
    if (t->fd < 0 || fstat(t->fd, &st) != 0) {
        sdice_set_error("sdice_junc_open: cannot open %s", path);
        sdice_junc_close(t);
        return SDICE_ERR_ARG;
    }
    t->size = (size_t)st.st_size;
    if (t->size) {
        void* m = mmap(nullptr, t->size, PROT_READ, MAP_PRIVATE, t->fd, 0);
        if (m == MAP_FAILED) {
            sdice_set_error("sdice_junc_open: mmap failed for %s", path);
            t->size = 0;
            sdice_junc_close(t);
            return SDICE_ERR_ARG;
        }
        t->base = (const char*)m;
    }
    size_t pos = 0;
    t->line_start.reserve(t->size / 32 + 2);
    while (pos < t->size) {
        t->line_start.push_back(pos);
        const char* e = (const char*)memchr(t->base + pos, '\n', t->size - pos);
        pos = e ? (size_t)(e + 1 - t->base) : t->size;
    }
    t->line_start.push_back(t->size);
    t->n = (int64_t)t->line_start.size() - 1;
    // chromosome table (serial: a handful of distinct names)
    t->chrom_id.resize((size_t)t->n);
    std::unordered_map<std::string, int32_t> ids;
    std::string last;
    int32_t last_id = -1;
    int64_t bytes = 0;
    for (int64_t i = 0; i < t->n; ++i) {
        const char* a = t->base + t->line_start[i];
        const char* b = t->base + t->line_start[i + 1];
        const char* tab = (const char*)memchr(a, '\t', (size_t)(b - a));
        const char* stop = tab ? tab : b;
        while (stop > a && (stop[-1] == '\n' || stop[-1] == '\r')) --stop;
        const size_t len = (size_t)(stop - a);
        if (last_id >= 0 && last.size() == len && memcmp(last.data(), a, len) == 0) {
            t->chrom_id[(size_t)i] = last_id;
            continue;
        }
        std::string name(a, len);
        auto it = ids.find(name);
        if (it == ids.end()) {
            it = ids.emplace(name, (int32_t)t->chroms.size()).first;
            t->chroms.push_back(name);
            bytes += (int64_t)len;
        }
        last = name;
        last_id = it->second;
        t->chrom_id[(size_t)i] = last_id;
    }
    if (n_lines) *n_lines = t->n;
    if (n_chroms) *n_chroms = (int32_t)t->chroms.size();
    if (chrom_bytes) *chrom_bytes = bytes;
    *out = t;
    return SDICE_OK;
} catch (const std::exception& e) {
    sdice_set_error("sdice_junc_open: %s", e.what());
    return SDICE_ERR_NOMEM;
} catch (...) {
    sdice_set_error("sdice_junc_open: unknown exception");
    return SDICE_ERR_STATE;
}

// Fills, per line: chrom_id (index into the file's chromosome table), left, right, strand code
// (0 '+', 1 '-', 2 anything else), score, admit (1 = passes the admission filters of its type).
// chrom_names / chrom_off[n_chroms+1]: the chromosome table.
extern "C" int sdice_junc_read(sdice_juncfile* t, int32_t min_length, int32_t max_length, int32_t min_unique,
                               int32_t min_overhang, double min_entropy, int no_multimap, int32_t* chrom_id,
                               int32_t* left, int32_t* right, int8_t* strand, int64_t* score, uint8_t* admit,
                               char* chrom_names, int64_t* chrom_off, int threads) try {
    if (!t || (t->n > 0 && (!chrom_id || !left || !right || !strand || !score || !admit))) {
        sdice_set_error("sdice_junc_read: bad arguments");
        return SDICE_ERR_ARG;
    }
    if (chrom_names && chrom_off) {
        int64_t off = 0;
        for (size_t c = 0; c < t->chroms.size(); ++c) {
            chrom_off[c] = off;
            memcpy(chrom_names + off, t->chroms[c].data(), t->chroms[c].size());
            off += (int64_t)t->chroms[c].size();
        }
        chrom_off[t->chroms.size()] = off;
    }
    int nthreads = threads > 0 ? threads : sd_default_threads();
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    if (t->n < 8192) nthreads = 1;
    std::vector<int64_t> bad((size_t)nthreads, -1);
    auto work = [&](int tix, int64_t a, int64_t b) {
        for (int64_t i = a; i < b; ++i) {
            const char* p = t->base + t->line_start[i];
            const char* end = t->base + t->line_start[i + 1];
            while (end > p && (end[-1] == '\n' || end[-1] == '\r' || end[-1] == ' ' || end[-1] == '\t')) --end;   // rstrip()
            chrom_id[i] = t->chrom_id[(size_t)i];
            Field f[9];
            int64_t l = 0, r = 0, sc = 0;
            bool ok;
            if (t->type == 2) {
                ok = split_tabs(p, end, f, 8);
                int64_t motif = 0, multi = 0;
                ok = ok && to_i64(f[1], l) && to_i64(f[2], r) && to_i64(f[4], motif) && to_i64(f[6], sc) && to_i64(f[7], multi);
                if (!ok) { bad[tix] = i; return; }
                l -= 1;
                if (!no_multimap) sc += multi;
                const size_t sl = (size_t)(f[3].b - f[3].a);
                int8_t st = 2;
                if (sl == 1 && (*f[3].a == '1' || *f[3].a == '+')) st = 0;
                else if (sl == 1 && (*f[3].a == '2' || *f[3].a == '-')) st = 1;
                else if (!(sl == 1 && *f[3].a == '0')) { bad[tix] = i; return; }     // KeyError in the reference
                const int64_t len = r - l;
                strand[i] = st;
                admit[i] = (len < max_length && len > min_length && st != 2 && sc >= min_unique && (motif == 1 || motif == 2)) ? 1 : 0;
            } else {
                ok = split_tabs(p, end, f, 6);
                ok = ok && to_i64(f[1], l) && to_i64(f[2], r) && to_i64(f[4], sc);
                if (!ok) { bad[tix] = i; return; }
                const size_t sl = (size_t)(f[5].b - f[5].a);
                int8_t st = 2;
                if (sl == 1 && *f[5].a == '+') st = 0;
                else if (sl == 1 && *f[5].a == '-') st = 1;
                strand[i] = st;
                const int64_t len = r - l;
                bool pass = true;
                if (t->type == 0) {
                    pass = sc >= min_unique && len <= max_length && len >= min_length;
                } else {
                    // name = e:<Lent>:<Rent>;o:<overhang>;m:<motif>;a:<annotation>
                    const char* s0 = f[3].a;
                    const char* se = f[3].b;
                    const char* semi[3];
                    int ns = 0;
                    for (const char* q = s0; q < se && ns < 3; ++q)
                        if (*q == ';') semi[ns++] = q;
                    if (ns < 3) { bad[tix] = i; return; }
                    // info[3][1]: text between the first ':' of the 4th item and the next ':' (or its end)
                    // (the 4th item ends at the next ';' if the name carries more items, SPLICEDICE.py:190)
                    const char* a4 = semi[2] + 1;
                    const char* a4e = (const char*)memchr(a4, ';', (size_t)(se - a4));
                    if (!a4e) a4e = se;
                    const char* c4 = (const char*)memchr(a4, ':', (size_t)(a4e - a4));
                    if (!c4) { bad[tix] = i; return; }
                    const char* v4 = c4 + 1;
                    const char* v4e = (const char*)memchr(v4, ':', (size_t)(a4e - v4));
                    if (!v4e) v4e = a4e;
                    const bool unannotated = (v4e - v4 == 1 && *v4 == '?');
                    if (unannotated) {
                        // info[0] = e:<Lent>:<Rent>, info[1] = o:<overhang>
                        const char* e0 = s0;
                        const char* e0e = semi[0];
                        const char* c1 = (const char*)memchr(e0, ':', (size_t)(e0e - e0));
                        const char* c2 = c1 ? (const char*)memchr(c1 + 1, ':', (size_t)(e0e - c1 - 1)) : nullptr;
                        const char* o0 = semi[0] + 1;
                        const char* o0e = semi[1];
                        const char* oc = (const char*)memchr(o0, ':', (size_t)(o0e - o0));
                        if (!c1 || !c2 || !oc) { bad[tix] = i; return; }
                        const char* c3 = (const char*)memchr(c2 + 1, ':', (size_t)(e0e - c2 - 1));
                        const char* oce = (const char*)memchr(oc + 1, ':', (size_t)(o0e - oc - 1));
                        double le = 0, re = 0;
                        int64_t ov = 0;
                        Field fo{oc + 1, oce ? oce : o0e};
                        if (!to_f64(c1 + 1, c2, le) || !to_f64(c2 + 1, c3 ? c3 : e0e, re) || !to_i64(fo, ov)) { bad[tix] = i; return; }
                        pass = !(sc < min_unique) && !(len > max_length || len < min_length) && !(ov < min_overhang) &&
                               !(le < min_entropy || re < min_entropy);
                    }
                }
                admit[i] = (pass && st != 2) ? 1 : 0;
            }
            if (l < INT32_MIN || l > INT32_MAX || r < INT32_MIN || r > INT32_MAX) { bad[tix] = i; return; }
            left[i] = (int32_t)l;
            right[i] = (int32_t)r;
            score[i] = sc;
        }
    };
    if (nthreads == 1) {
        work(0, 0, t->n);
    } else {
        std::vector<std::thread> pool;
        JoinAll joiner{pool};          // (a thread that cannot be started throws: the started ones are joined first)
        for (int k = 0; k < nthreads; ++k) pool.emplace_back(work, k, t->n * k / nthreads, t->n * (k + 1) / nthreads);
    }
    int64_t first_bad = -1;
    for (auto b : bad)
        if (b >= 0 && (first_bad < 0 || b < first_bad)) first_bad = b;
    if (first_bad >= 0) {
        sdice_set_error("sdice_junc_read: malformed line %lld", (long long)(first_bad + 1));
        return SDICE_ERR_ARG;
    }
    return SDICE_OK;
} catch (const std::exception& e) {
    sdice_set_error("sdice_junc_read: %s", e.what());
    return SDICE_ERR_NOMEM;
} catch (...) {
    sdice_set_error("sdice_junc_read: unknown exception");
    return SDICE_ERR_STATE;
}

// Row of every query junction in a table of rows sorted by (chrom, left, right, strand); -1 if absent.
extern "C" int sdice_junc_lookup(int64_t n_rows, const int32_t* row_chrom, const int32_t* row_left,
                                 const int32_t* row_right, const int8_t* row_strand, int64_t n_q,
                                 const int32_t* q_chrom, const int32_t* q_left, const int32_t* q_right,
                                 const int8_t* q_strand, int32_t* row_out, int threads) try {
    if (n_rows < 0 || n_q < 0 || (n_q > 0 && !row_out)) { sdice_set_error("sdice_junc_lookup: bad arguments"); return SDICE_ERR_ARG; }
    int nthreads = threads > 0 ? threads : sd_default_threads();
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    if (n_q < 8192) nthreads = 1;
    auto less = [&](int64_t r, int32_t c, int32_t l, int32_t rr, int8_t s) {
        if (row_chrom[r] != c) return row_chrom[r] < c;
        if (row_left[r] != l) return row_left[r] < l;
        if (row_right[r] != rr) return row_right[r] < rr;
        return row_strand[r] < s;
    };
    auto work = [&](int64_t a, int64_t b) {
        for (int64_t i = a; i < b; ++i) {
            const int32_t c = q_chrom[i], l = q_left[i], rr = q_right[i];
            const int8_t s = q_strand[i];
            int64_t lo = 0, hi = n_rows;
            while (lo < hi) {
                const int64_t mid = lo + ((hi - lo) >> 1);
                if (less(mid, c, l, rr, s)) lo = mid + 1; else hi = mid;
            }
            row_out[i] = (lo < n_rows && c >= 0 && row_chrom[lo] == c && row_left[lo] == l && row_right[lo] == rr &&
                          row_strand[lo] == s) ? (int32_t)lo : -1;
        }
    };
    if (nthreads == 1) {
        work(0, n_q);
    } else {
        std::vector<std::thread> pool;
        JoinAll joiner{pool};
        for (int k = 0; k < nthreads; ++k) pool.emplace_back(work, n_q * k / nthreads, n_q * (k + 1) / nthreads);
    }
    return SDICE_OK;
} catch (const std::exception& e) {
    sdice_set_error("sdice_junc_lookup: %s", e.what());
    return SDICE_ERR_NOMEM;
} catch (...) {
    sdice_set_error("sdice_junc_lookup: unknown exception");
    return SDICE_ERR_STATE;
}

// One sample's column of the count table (SPLICEDICE.getJunctionCounts, SPLICEDICE.py:257-295: counts[junction][sample] =
// score; a later line of the file overwrites an earlier one, and no score filter applies): col[row] = score for every
// line whose junction is among the rows, in line order.  `col` is the sample's int32 [n_rows] stretch of the TRANSPOSED
// table, zeroed by the caller -- a sample writes 4 n_rows contiguous bytes instead of one word in each of n_rows lines of
// the [row][sample] table -- and sdice_transpose_i32 turns the finished table.  low (or NULL): set to 1 where a line with
// score < min_unique hits the row (it stays set when a later line raises the count).  A FINAL value outside [0, 2^31) is
// an error.  Single-threaded: the caller runs one call per sample side by side.
extern "C" int sdice_junc_count_column(int64_t n_rows, const int32_t* row_chrom, const int32_t* row_left,
                                       const int32_t* row_right, const int8_t* row_strand, int64_t n_q,
                                       const int32_t* q_chrom, const int32_t* q_left, const int32_t* q_right,
                                       const int8_t* q_strand, const int64_t* score, int32_t min_unique, int32_t* col,
                                       uint8_t* low) try {
    if (n_rows < 0 || n_q < 0 || (n_q > 0 && (!q_chrom || !q_left || !q_right || !q_strand || !score)) || (n_rows > 0 && !col)) {
        sdice_set_error("sdice_junc_count_column: bad arguments");
        return SDICE_ERR_ARG;
    }
    constexpr int32_t BAD = INT32_MIN;          // (a value that no admissible count takes) marks a row whose last value is out of range
    int64_t n_bad = 0;
    int64_t hint = 0;                           // files are sorted by coordinate as a rule: the next junction is the next row or near it
    for (int64_t i = 0; i < n_q; ++i) {
        const int32_t c = q_chrom[i], l = q_left[i], rr = q_right[i];
        const int8_t st = q_strand[i];
        auto cmp = [&](int64_t r) {             // rows[r] <=> query: -1, 0, 1
            if (row_chrom[r] != c) return row_chrom[r] < c ? -1 : 1;
            if (row_left[r] != l) return row_left[r] < l ? -1 : 1;
            if (row_right[r] != rr) return row_right[r] < rr ? -1 : 1;
            if (row_strand[r] != st) return row_strand[r] < st ? -1 : 1;
            return 0;
        };
        int64_t lo = 0, hi = n_rows;
        if (hint < n_rows) {                    // gallop from the previous hit
            const int h = cmp(hint);
            if (h == 0) { lo = hint; hi = hint; }
            else if (h < 0) {
                int64_t step = 1, a = hint + 1;
                while (a < n_rows && cmp(a) < 0) { lo = a + 1; a += step; step <<= 1; }
                if (lo < hint + 1) lo = hint + 1;
                hi = a < n_rows ? a : n_rows;
            } else {
                hi = hint;
            }
        }
        while (lo < hi) {
            const int64_t mid = lo + ((hi - lo) >> 1);
            if (cmp(mid) < 0) lo = mid + 1; else hi = mid;
        }
        if (lo >= n_rows || c < 0 || cmp(lo) != 0) continue;
        hint = lo + 1;
        const int64_t v = score[i];
        if (col[lo] == BAD) --n_bad;
        if (v < 0 || v > (int64_t)INT32_MAX) { col[lo] = BAD; ++n_bad; }
        else col[lo] = (int32_t)v;
        if (low && v < (int64_t)min_unique) low[lo] = 1;
    }
    if (n_bad > 0) {
        sdice_set_error("junction counts must be non-negative and below 2**31");
        return SDICE_ERR_ARG;
    }
    return SDICE_OK;
} catch (const std::exception& e) {
    sdice_set_error("sdice_junc_count_column: %s", e.what());
    return SDICE_ERR_NOMEM;
} catch (...) {
    sdice_set_error("sdice_junc_count_column: unknown exception");
    return SDICE_ERR_STATE;
}

// dst[c][r] = src[r][c] for an int32 table of `rows` x `cols` (blocks of 64 x 64 words, threads over the blocks of rows)
extern "C" int sdice_transpose_i32(int64_t rows, int64_t cols, const int32_t* src, int32_t* dst, int threads) try {
    if (rows < 0 || cols < 0 || (rows > 0 && cols > 0 && (!src || !dst))) { sdice_set_error("sdice_transpose_i32: bad arguments"); return SDICE_ERR_ARG; }
    int nthreads = threads > 0 ? threads : sd_default_threads();
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    constexpr int64_t BL = 64;
    const int64_t cblocks = (cols + BL - 1) / BL;
    if (rows * cols < (int64_t)1 << 20 || cblocks < nthreads) nthreads = 1;
    auto work = [&](int64_t cb0, int64_t cb1) {          // a thread owns whole stretches of dst rows
        for (int64_t cb = cb0; cb < cb1; ++cb) {
            const int64_t c0 = cb * BL, c1 = std::min(cols, c0 + BL);
            for (int64_t r0 = 0; r0 < rows; r0 += BL) {
                const int64_t r1 = std::min(rows, r0 + BL);
                for (int64_t c = c0; c < c1; ++c) {
                    int32_t* d = dst + c * rows;
                    for (int64_t r = r0; r < r1; ++r) d[r] = src[r * cols + c];
                }
            }
        }
    };
    if (nthreads == 1) {
        work(0, cblocks);
    } else {
        std::vector<std::thread> pool;
        JoinAll joiner{pool};
        for (int k = 0; k < nthreads; ++k) pool.emplace_back(work, cblocks * k / nthreads, cblocks * (k + 1) / nthreads);
    }
    return SDICE_OK;
} catch (const std::exception& e) {
    sdice_set_error("sdice_transpose_i32: %s", e.what());
    return SDICE_ERR_NOMEM;
} catch (...) {
    sdice_set_error("sdice_transpose_i32: unknown exception");
    return SDICE_ERR_STATE;
}

// worker threads the host-side calls use by default: hardware threads capped by the cgroup's CPU quota
extern "C" int sdice_host_threads(void) { return sd_default_threads(); }

// Second step of the ingest, per file (SPLICEDICE.getAllJunctions adds (chrom, left, right, strand) to one set and sorts
// it, SPLICEDICE.py:147-228): chrom_rank[i] = rank_of_chrom[chrom_id[i]] (the file's chromosome table mapped to the ranks of
// the names in the sorted union of all files' tables), and the admitted lines' keys in the order-preserving packing
// chrom 12 | left 31 | right - left 20 | strand 1 bits, compacted into keys_out (room for n).  *packable = 0 when an
// admitted junction does not fit the packing (the caller then sorts tuples on the host); *n_keys = keys written.
extern "C" int sdice_junc_pack_keys(int64_t n, const int32_t* chrom_id, const int32_t* rank_of_chrom, int32_t n_chroms,
                                    const int32_t* left, const int32_t* right, const int8_t* strand, const uint8_t* admit,
                                    int32_t* chrom_rank, uint64_t* keys_out, int64_t* n_keys, int32_t* packable) try {
    if (n < 0 || !n_keys || !packable || (n > 0 && (!chrom_id || !rank_of_chrom || !left || !right || !strand || !admit || !chrom_rank || !keys_out))) {
        sdice_set_error("sdice_junc_pack_keys: bad arguments");
        return SDICE_ERR_ARG;
    }
    int64_t k = 0;
    bool fits = true;
    for (int64_t i = 0; i < n; ++i) {
        const int32_t ci = chrom_id[i];
        if (ci < 0 || ci >= n_chroms) { sdice_set_error("sdice_junc_pack_keys: chromosome index out of range"); return SDICE_ERR_ARG; }
        const int32_t c = rank_of_chrom[ci];
        chrom_rank[i] = c;
        if (!admit[i]) continue;
        const int64_t l = left[i], span = (int64_t)right[i] - l;
        if (c < 0 || c >= (1 << 12) || l < 0 || span < 0 || span >= ((int64_t)1 << 20)) { fits = false; continue; }
        keys_out[k++] = ((uint64_t)c << 52) | ((uint64_t)l << 21) | ((uint64_t)span << 1) | (uint64_t)(strand[i] & 1);
    }
    *n_keys = k;
    *packable = fits ? 1 : 0;
    return SDICE_OK;
} catch (const std::exception& e) {
    sdice_set_error("sdice_junc_pack_keys: %s", e.what());
    return SDICE_ERR_NOMEM;
} catch (...) {
    sdice_set_error("sdice_junc_pack_keys: unknown exception");
    return SDICE_ERR_STATE;
}
