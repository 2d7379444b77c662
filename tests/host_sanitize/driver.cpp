// Sanitizer driver for the threaded host C++ of the library (csrc/textio.cpp, csrc/juncio.cpp):
// built by `make -C splicedice_amd/csrc asan` / `tsan` together with those two files (no HIP, no GPU)
// and run by tests/test_host_sanitizers.py.  It pushes every host entry point through its threaded
// path: table writer (3 modes) -> table reader round trip, column writer, cluster writer, junction
// parser on the golden inputs (all file types) and the row lookup.  Any ASan / UBSan / TSan report
// makes the process exit non-zero.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <string>
#include <thread>
#include <vector>
#include "sdice.h"

static char g_err[1024];
void sdice_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
}
#define CHECK(x) do { if (!(x)) { fprintf(stderr, "FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #x, g_err); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: driver <tmpdir> <junction file>:<type> ...\n"); return 2; }
    const std::string tmp = argv[1];
    const int64_t n = 20011; const int s = 37;
    std::vector<std::string> nm(n);
    std::string names; std::vector<int64_t> off(n + 1, 0);
    for (int64_t i = 0; i < n; ++i) { nm[i] = "chr" + std::to_string(i % 23) + ":" + std::to_string(i * 7) + "-" + std::to_string(i * 7 + 100) + ":+"; names += nm[i]; off[i + 1] = (int64_t)names.size(); }
    std::vector<float> f32((size_t)n * s); std::vector<double> f64((size_t)n * s); std::vector<int32_t> i32((size_t)n * s);
    unsigned x = 12345u;
    for (size_t k = 0; k < f32.size(); ++k) {
        x = x * 1664525u + 1013904223u;
        f32[k] = (x >> 8) % 97 == 0 ? NAN : (float)((x >> 9) % 1001) / 1000.0f;
        f64[k] = (double)(x >> 3) / 536870912.0 * ((x & 7) == 0 ? 1e-30 : 1.0);
        i32[k] = (int32_t)((x >> 7) % 5000);
    }
    std::string hdr = "cluster";
    for (int c = 0; c < s; ++c) hdr += "\tS" + std::to_string(c);
    hdr += "\n";
    const std::string p3 = tmp + "/t3.tsv", p0 = tmp + "/t0.tsv", pr = tmp + "/tr.tsv", pc = tmp + "/tc.tsv", pk = tmp + "/tk.tsv";
    for (int threads : {1, 0, 7}) {
        CHECK(sdice_write_table(p3.c_str(), hdr.c_str(), n, s, names.data(), off.data(), f32.data(), 0, 0, threads) == 0);
        CHECK(sdice_write_table(p0.c_str(), hdr.c_str(), n, s, names.data(), off.data(), i32.data(), 2, 1, threads) == 0);
        CHECK(sdice_write_table(pr.c_str(), hdr.c_str(), n, s, names.data(), off.data(), f64.data(), 1, 2, threads) == 0);
        // round trip: '.3f' text of k/1000 values reads back as the same float32; repr text of float64 is exact
        sdice_table* t = nullptr; int64_t rn = 0, nb = 0, hb = 0; int32_t rs = 0;
        CHECK(sdice_table_open(p3.c_str(), &t, &rn, &rs, &nb, &hb) == 0 && rn == n && rs == s);
        std::vector<char> rh((size_t)hb + 1), rnames((size_t)nb + 1); std::vector<int64_t> roff(n + 1); std::vector<float> back((size_t)n * s);
        CHECK(sdice_table_read(t, rh.data(), rnames.data(), roff.data(), back.data(), 0, threads) == 0);
        CHECK(sdice_table_close(t) == 0);
        for (size_t k = 0; k < back.size(); ++k) CHECK((isnan(back[k]) && isnan(f32[k])) || back[k] == f32[k]);
        CHECK(roff[n] == off[n] && memcmp(rnames.data(), names.data(), names.size()) == 0);
        CHECK(sdice_table_open(pr.c_str(), &t, &rn, &rs, &nb, &hb) == 0);
        std::vector<double> back64((size_t)n * s);
        rh.assign((size_t)hb + 1, 0); rnames.assign((size_t)nb + 1, 0);
        CHECK(sdice_table_read(t, rh.data(), rnames.data(), roff.data(), back64.data(), 1, threads) == 0);
        CHECK(sdice_table_close(t) == 0);
        for (size_t k = 0; k < back64.size(); ++k) CHECK(back64[k] == f64[k]);
        // column writer: mixed dtypes / modes
        std::vector<float> c0(n); std::vector<double> c1(n);
        for (int64_t i = 0; i < n; ++i) { c0[i] = f32[(size_t)i * s]; c1[i] = f64[(size_t)i * s]; }
        const void* cols[2] = {c0.data(), c1.data()}; const int32_t dt[2] = {0, 1}, md[2] = {2, 2};
        CHECK(sdice_write_columns(pc.c_str(), "event\tx\ty\n", n, names.data(), off.data(), 2, cols, dt, md, threads) == 0);
        // cluster writer: a ring of neighbours + an empty list
        std::vector<int64_t> rp(n + 1, 0); std::vector<int32_t> col;
        for (int64_t i = 0; i < n; ++i) { if (i % 5) { col.push_back((int32_t)((i + 1) % n)); col.push_back((int32_t)((i + n - 1) % n)); } rp[i + 1] = (int64_t)col.size(); }
        CHECK(sdice_write_clusters(pk.c_str(), n, names.data(), off.data(), rp.data(), col.data(), threads) == 0);
        col[3] = (int32_t)n + 5;                     // an out-of-range neighbour must be refused, not read
        CHECK(sdice_write_clusters(pk.c_str(), n, names.data(), off.data(), rp.data(), col.data(), threads) != 0);
        // junction bed writer: three chromosome names, an out-of-range chromosome index must be refused
        const std::string cn = "chr1chrXKI270728.1"; const int64_t co[4] = {0, 4, 8, 18};
        std::vector<int32_t> ch(n), lf(n), rt(n); std::string sd((size_t)n, '+');
        for (int64_t i = 0; i < n; ++i) { ch[i] = (int32_t)(i % 3); lf[i] = (int32_t)(i * 7); rt[i] = (int32_t)(i * 7 + 100 + i % 11); if (i % 2) sd[i] = '-'; }
        CHECK(sdice_write_junction_bed(pk.c_str(), n, cn.data(), co, 3, ch.data(), lf.data(), rt.data(), sd.data(), threads) == 0);
        if (n > 2) { ch[2] = 3; CHECK(sdice_write_junction_bed(pk.c_str(), n, cn.data(), co, 3, ch.data(), lf.data(), rt.data(), sd.data(), threads) != 0); }
    }
    // junction parser + lookup on every file given as path:type
    for (int a = 2; a < argc; ++a) {
        std::string arg = argv[a];
        const size_t c = arg.rfind(':');
        const std::string path = arg.substr(0, c); const int type = atoi(arg.c_str() + c + 1);
        for (int threads : {1, 0, 5}) {
            sdice_juncfile* f = nullptr; int64_t nl = 0, cb = 0; int32_t nc = 0;
            CHECK(sdice_junc_open(path.c_str(), type, &f, &nl, &nc, &cb) == 0);
            std::vector<int32_t> ci(nl + 1), l(nl + 1), r(nl + 1); std::vector<int8_t> st(nl + 1); std::vector<int64_t> sc(nl + 1), coff(nc + 1);
            std::vector<uint8_t> ad(nl + 1); std::vector<char> cn((size_t)cb + 1);
            CHECK(sdice_junc_read(f, 50, 50000, 5, 5, 1.0, 0, ci.data(), l.data(), r.data(), st.data(), sc.data(), ad.data(), cn.data(), coff.data(), threads) == 0);
            CHECK(sdice_junc_close(f) == 0);
            int64_t admitted = 0;
            for (int64_t i = 0; i < nl; ++i) admitted += ad[i];
            CHECK(nl > 0 && admitted > 0);
        }
    }
    {   // row lookup: every row finds itself, a stranger does not
        std::vector<int32_t> rc(n), rl(n), rr(n), out(n + 1); std::vector<int8_t> rs8(n);
        for (int64_t i = 0; i < n; ++i) { rc[i] = (int32_t)(i / 1000); rl[i] = (int32_t)(i % 1000) * 10; rr[i] = rl[i] + 5; rs8[i] = 0; }
        CHECK(sdice_junc_lookup(n, rc.data(), rl.data(), rr.data(), rs8.data(), n, rc.data(), rl.data(), rr.data(), rs8.data(), out.data(), 0) == 0);
        for (int64_t i = 0; i < n; ++i) CHECK(out[i] == (int32_t)i);
        int32_t qc = 3, ql = 7, qr = 8; int8_t qs = 1; int32_t o1 = 0;
        CHECK(sdice_junc_lookup(n, rc.data(), rl.data(), rr.data(), rs8.data(), 1, &qc, &ql, &qr, &qs, &o1, 1) == 0 && o1 == -1);
    }
    {   // count columns side by side + transpose, key packing, row names (the ingest's second half)
        const int64_t nr = 20000; const int32_t ns = 6;
        std::vector<int32_t> rc(nr), rl(nr), rr(nr); std::vector<int8_t> rs8(nr);
        for (int64_t i = 0; i < nr; ++i) { rc[i] = (int32_t)(i / 5000); rl[i] = (int32_t)(i % 5000) * 10; rr[i] = rl[i] + 7; rs8[i] = (int8_t)(i & 1); }
        std::vector<int32_t> table_t((size_t)ns * nr, 0), table((size_t)ns * nr, -1);
        std::vector<uint8_t> low((size_t)ns * nr, 0);
        std::vector<int64_t> score(nr);
        for (int64_t i = 0; i < nr; ++i) score[i] = i % 9;
        std::vector<std::thread> pool;
        std::atomic<int> bad{0};
        for (int32_t c = 0; c < ns; ++c)
            pool.emplace_back([&, c] {
                if (sdice_junc_count_column(nr, rc.data(), rl.data(), rr.data(), rs8.data(), nr, rc.data(), rl.data(), rr.data(), rs8.data(),
                                            score.data(), 5, table_t.data() + (size_t)c * nr, low.data() + (size_t)c * nr) != 0) bad++;
            });
        for (auto& th : pool) th.join();
        CHECK(bad == 0);
        CHECK(sdice_transpose_i32(ns, nr, table_t.data(), table.data(), 4) == 0);
        for (int64_t i = 0; i < nr; i += 997) for (int32_t c = 0; c < ns; ++c) CHECK(table[(size_t)i * ns + c] == (int32_t)(i % 9) && low[(size_t)c * nr + i] == (i % 9 < 5));
        int32_t rank_of[4] = {2, 0, 3, 1};
        std::vector<int32_t> crank(nr); std::vector<uint64_t> keys(nr); std::vector<uint8_t> admit(nr, 1);
        int64_t nk = 0; int32_t packable = 0;
        CHECK(sdice_junc_pack_keys(nr, rc.data(), rank_of, 4, rl.data(), rr.data(), rs8.data(), admit.data(), crank.data(), keys.data(), &nk, &packable) == 0);
        CHECK(nk == nr && packable == 1 && crank[5000] == 0 && (keys[1] >> 52) == 2);
        const char cnames[] = "chrAchrBBchrCchrDDD"; int64_t coff[5] = {0, 4, 9, 13, 19};
        std::vector<char> strand(nr); for (int64_t i = 0; i < nr; ++i) strand[i] = rs8[i] ? '-' : '+';
        std::vector<char> nm((size_t)nr * 40); std::vector<int64_t> noff(nr + 1); int64_t need = 0;
        CHECK(sdice_junction_names(nr, cnames, coff, 4, rc.data(), rl.data(), rr.data(), strand.data(), nm.data(), (int64_t)nm.size(), noff.data(), &need) == 0);
        CHECK(need == noff[nr] && std::string(nm.data() + noff[1], (size_t)(noff[2] - noff[1])) == "chrA:10-17:-");
        CHECK(sdice_junction_names(nr, cnames, coff, 4, rc.data(), rl.data(), rr.data(), strand.data(), nm.data(), 10, noff.data(), &need) != 0);
        CHECK(sdice_host_threads() >= 1 && sdice_textio_trim() == 0);
    }
    printf("host sanitizer driver: ok\n");
    return 0;
}
