// host_tail.cpp -- measurement tooling (not product): the HOST work the Rust shim (integration/p2hot.rs) does after
// p2hot_commit_salted returns, timed on the box's own cores so that bench.py can price the drop-in's default mode next to the
// GPU call (`other_configs.host_c3_wires_leaves_back`).  It mirrors the shim's code paths one for one:
//   split_coeffs   `coeffs.par_chunks_exact(n).map(|c| PolynomialCoeffs::new(c.to_vec())).collect()`: W allocations + 8 W n bytes
//                  copied, in parallel -- ALL the default mode (P2HOT_LEAVES=host) does: the leaf matrix stays the one flat
//                  buffer the library filled, behind MerkleTree::get
//   rows_parallel  P2HOT_LEAVES=vec: `flat.par_chunks_exact(w).map(|r| r.to_vec()).collect()`: N allocations + 8 W N bytes
//   rows_serial    the round-3 shim: the same on ONE thread (8.4 M allocations + 9 GB per wires commitment at 2^20 gates)
// The two row modes run on a sample of 2^sample_log rows and are scaled to N (they are linear in the row count).
// build: g++ -O2 -std=c++17 -pthread -o tools/host_tail tools/host_tail.cpp      usage: host_tail W log_n rate_bits [sample_log] [threads]
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <thread>
#include <vector>

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// the cores this job may use: cgroup v2 cpu.max (quota / period) caps hardware_concurrency
static unsigned usable_cores() {
    unsigned hc = std::max(1u, std::thread::hardware_concurrency());
    std::ifstream f("/sys/fs/cgroup/cpu.max");
    std::string q;
    double period = 0;
    if (f >> q >> period && q != "max" && period > 0) {
        const double c = atof(q.c_str()) / period;
        if (c >= 1) hc = std::min(hc, (unsigned)(c + 0.5));
    }
    return hc;
}

template <class F>
static void parallel_for(size_t count, unsigned threads, F body) {  // contiguous index ranges, like rayon's chunked split
    std::vector<std::thread> th;
    for (unsigned t = 0; t < threads; ++t)
        th.emplace_back([=] {
            const size_t lo = count * t / threads, hi = count * (t + 1) / threads;
            for (size_t i = lo; i < hi; ++i) body(i);
        });
    for (auto &t : th) t.join();
}

int main(int argc, char **argv) {
    const size_t W = argc > 1 ? strtoull(argv[1], nullptr, 10) : 135;
    const unsigned log_n = argc > 2 ? atoi(argv[2]) : 20, rate_bits = argc > 3 ? atoi(argv[3]) : 3;
    unsigned sample_log = argc > 4 ? atoi(argv[4]) : 21;
    const unsigned threads = argc > 5 ? std::max(1, atoi(argv[5])) : usable_cores();
    const size_t n = (size_t)1 << log_n, N = n << rate_bits;
    sample_log = std::min(sample_log, log_n + rate_bits);
    const size_t rows = (size_t)1 << sample_log;
    // ---- split_coeffs: the flat [W][n] coefficient block -> W vectors
    std::vector<uint64_t> flat(W * n);
    for (size_t i = 0; i < flat.size(); i += 512) flat[i] = i;  // touched: the library's copy has just written it
    double split_ms = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        std::vector<std::vector<uint64_t>> polys(W);
        const double t0 = now_ms();
        parallel_for(W, std::min<unsigned>(threads, (unsigned)W), [&](size_t c) { polys[c].assign(flat.begin() + c * n, flat.begin() + (c + 1) * n); });
        split_ms = std::min(split_ms, now_ms() - t0);
        if (polys[W - 1][n - 1] != flat[W * n - 1]) return 1;
    }
    std::vector<uint64_t>().swap(flat);
    // ---- rows: the flat [rows][W] leaf block -> one vector per row
    std::vector<uint64_t> leaves(rows * W);
    for (size_t i = 0; i < leaves.size(); i += 512) leaves[i] = i;
    double par_ms = 1e30, ser_ms = 1e30;
    for (int rep = 0; rep < 2; ++rep) {
        std::vector<std::vector<uint64_t>> v(rows);
        const double t0 = now_ms();
        parallel_for(rows, threads, [&](size_t r) { v[r].assign(leaves.begin() + r * W, leaves.begin() + (r + 1) * W); });
        par_ms = std::min(par_ms, now_ms() - t0);
        if (v[rows - 1][W - 1] != leaves[rows * W - 1]) return 1;
    }
    {
        const size_t srows = std::min(rows, (size_t)1 << 20);  // the serial loop is slow: a 2^20-row sample
        std::vector<std::vector<uint64_t>> v(srows);
        const double t0 = now_ms();
        for (size_t r = 0; r < srows; ++r) v[r].assign(leaves.begin() + r * W, leaves.begin() + (r + 1) * W);
        ser_ms = (now_ms() - t0) * (double)rows / (double)srows;
    }
    const double scale = (double)N / (double)rows;
    printf("{\"W\": %zu, \"log_n\": %u, \"rate_bits\": %u, \"threads\": %u, \"split_coeffs_ms\": %.3f, \"rows_sample_log\": %u, "
           "\"rows_parallel_ms\": %.1f, \"rows_serial_ms\": %.1f, \"coeff_bytes\": %zu, \"leaf_bytes\": %zu}\n",
           W, log_n, rate_bits, threads, split_ms, sample_log, par_ms * scale, ser_ms * scale, W * n * 8, W * N * 8);
    return 0;
}
