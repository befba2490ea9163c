// sgx_testbench -- a C++ host for libsgx.so in the role of the reference's HLS testbench
// (gnn-rfsoc-mt-all-2022/src/main_float.cpp: load the CSR text matrices, call kernelmult1, print
// rows of D).  It binds nothing but include/sgx.h and the HIP runtime: the drop-in boundary as a
// C/C++ caller sees it.
//
//   sgx_testbench --adj A.txt --fea X.txt --weights W.txt --p P [--gemm-mode 0|1] [--relu 0|1]
//                 [--dtype f16|f32] [--exact] [--spmm-block S] [--fea-threads T] [--adj-threads T]
//                 [--rows 0,31] [--cols C] [--time ITERS]
//
// File formats are the reference's (main_float.cpp:415-536, :149-200): a CSR matrix is three
// comma-separated lines rowptr / colidx / values; with --gemm-mode 1 the feature file holds
// M_adj lines of M_fea values; the weight file holds M_fea lines of at least P values.
// Output: the testbench's own line format (main_float.cpp:358), e.g.
//   out :data index= 0 10 kernel = 0.0995483
// so that a run on the citeseer matrices with --exact --spmm-block 4 can be laid beside
// hls/.../csim/report/mmult_top_csim.log.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "sgx.h"

namespace {

typedef _Float16 half_t;

[[noreturn]] void die(const std::string &msg)
{
    std::cerr << "sgx_testbench: " << msg << std::endl;
    std::exit(2);
}

#define HIP_OK(expr)                                                          \
    do {                                                                      \
        hipError_t e_ = (expr);                                               \
        if (e_ != hipSuccess) die(std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

#define SGX_CALL(expr)                                                        \
    do {                                                                      \
        int s_ = (expr);                                                      \
        if (s_ != SGX_OK) die(std::string(#expr) + ": " + sgx_status_string(s_)); \
    } while (0)

// every number on one line, separators = commas and blanks; text -> float as `stream >> float` does
std::vector<float> numbers(const std::string &line)
{
    std::vector<float> out;
    const char *p = line.c_str();
    while (*p) {
        while (*p == ',' || *p == ' ' || *p == '\t' || *p == '\r') ++p;
        if (!*p) break;
        char *end = nullptr;
        const float v = std::strtof(p, &end);
        if (end == p) die("cannot parse a number near '" + std::string(p).substr(0, 20) + "'");
        out.push_back(v);
        p = end;
    }
    return out;
}

std::vector<std::string> lines_of(const std::string &path)
{
    std::ifstream f(path);
    if (!f.is_open()) die("cannot open " + path);
    std::vector<std::string> lines;
    std::string ln;
    while (std::getline(f, ln))
        if (ln.find_first_not_of(" \t\r,") != std::string::npos) lines.push_back(ln);
    return lines;
}

struct Csr {
    std::vector<int32_t> rowptr, col;
    std::vector<float> val;
};

Csr read_csr(const std::string &path)
{
    const std::vector<std::string> ln = lines_of(path);
    if (ln.size() < 3) die(path + ": expected three lines (rowptr / colidx / values)");
    Csr m;
    for (float v : numbers(ln[0])) m.rowptr.push_back((int32_t)v);
    for (float v : numbers(ln[1])) m.col.push_back((int32_t)v);
    m.val = numbers(ln[2]);
    if (m.rowptr.empty() || m.col.size() != m.val.size() || (size_t)m.rowptr.back() != m.col.size())
        die(path + ": the three lines do not describe one CSR matrix");
    return m;
}

template <typename T> std::vector<T> narrow(const std::vector<float> &v)
{
    std::vector<T> out(v.size());
    for (size_t i = 0; i < v.size(); ++i) out[i] = (T)v[i];      // float -> half: round to nearest even
    return out;
}

template <typename T> T *upload(const std::vector<T> &host)
{
    T *dev = nullptr;
    HIP_OK(hipMalloc(&dev, host.empty() ? 16 : host.size() * sizeof(T)));
    if (!host.empty()) HIP_OK(hipMemcpy(dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return dev;
}

struct Args {
    std::string adj, fea, weights, rows = "0";
    int p = 0, gemm_mode = 0, relu = 0, exact = 0, spmm_block = 1, fea_threads = 1, adj_threads = 1, cols = -1, time = 0;
    bool f32 = false;
};

Args parse(int argc, char **argv)
{
    Args a;
    for (int i = 1; i < argc; ++i) {
        const std::string k = argv[i];
        auto next = [&]() -> std::string {
            if (i + 1 >= argc) die("missing value after " + k);
            return argv[++i];
        };
        if (k == "--adj") a.adj = next();
        else if (k == "--fea") a.fea = next();
        else if (k == "--weights") a.weights = next();
        else if (k == "--p") a.p = std::atoi(next().c_str());
        else if (k == "--gemm-mode") a.gemm_mode = std::atoi(next().c_str());
        else if (k == "--relu") a.relu = std::atoi(next().c_str());
        else if (k == "--dtype") a.f32 = next() == "f32";
        else if (k == "--exact") a.exact = 1;
        else if (k == "--spmm-block") a.spmm_block = std::atoi(next().c_str());
        else if (k == "--fea-threads") a.fea_threads = std::atoi(next().c_str());
        else if (k == "--adj-threads") a.adj_threads = std::atoi(next().c_str());
        else if (k == "--rows") a.rows = next();
        else if (k == "--cols") a.cols = std::atoi(next().c_str());
        else if (k == "--time") a.time = std::atoi(next().c_str());
        else die("unknown option " + k);
    }
    if (a.adj.empty() || a.fea.empty() || a.weights.empty() || a.p < 1) die("--adj, --fea, --weights and --p are required");
    return a;
}

template <typename T> int run(const Args &a)
{
    const Csr A = read_csr(a.adj);
    const int N = (int)A.rowptr.size() - 1;

    // weights: M_fea lines of >= P numbers -> B = W^T [P][M_fea] (the layout the kernel reads, K.cpp:3043)
    const std::vector<std::string> wl = lines_of(a.weights);
    const int M_fea = (int)wl.size();
    std::vector<float> Bt((size_t)a.p * M_fea);
    for (int i = 0; i < M_fea; ++i) {
        const std::vector<float> row = numbers(wl[i]);
        if ((int)row.size() < a.p) die(a.weights + ": line " + std::to_string(i) + " holds fewer than P values");
        for (int j = 0; j < a.p; ++j) Bt[(size_t)j * M_fea + i] = row[j];
    }

    sgx_layer_desc d;
    std::memset(&d, 0, sizeof d);
    d.gemm_mode = a.gemm_mode; d.relu = a.relu;
    d.N_adj = N; d.M_adj = N; d.M_fea = M_fea; d.P_w = a.p;
    d.dtype = a.f32 ? SGX_F32 : SGX_F16;
    d.acc_mode = a.exact ? SGX_ACC_REF_HALF : SGX_ACC_F32;
    d.spmm_block = a.spmm_block; d.fea_threads = a.fea_threads; d.adj_threads = a.adj_threads;
    d.alpha = 0.2f;

    d.rowPtr_adj = upload(A.rowptr);
    d.columnIndex_adj = upload(A.col);
    d.values_adj = upload(narrow<T>(A.val));
    d.B = upload(narrow<T>(Bt));
    if (a.gemm_mode == 0) {
        const Csr X = read_csr(a.fea);
        if ((int)X.rowptr.size() - 1 != N) die("feature matrix and adjacency disagree on the number of nodes");
        d.rowPtr_fea = upload(X.rowptr);
        d.columnIndex_fea = upload(X.col);
        d.values_fea = upload(narrow<T>(X.val));
    } else {
        std::vector<float> dense;
        for (const std::string &ln : lines_of(a.fea)) {
            const std::vector<float> row = numbers(ln);
            if ((int)row.size() != M_fea) die(a.fea + ": a dense feature row does not hold M_fea values");
            dense.insert(dense.end(), row.begin(), row.end());
        }
        if ((int)(dense.size() / M_fea) != N) die("dense feature matrix and adjacency disagree on the number of nodes");
        d.values_fea = upload(narrow<T>(dense));
    }
    T *D = nullptr;
    HIP_OK(hipMalloc(&D, (size_t)N * a.p * sizeof(T)));
    d.D = D;
    d.workspace_bytes = sgx_layer_workspace_bytes(&d);
    if (d.workspace_bytes == 0) die("sgx_layer_workspace_bytes rejected the description");
    HIP_OK(hipMalloc(&d.workspace, d.workspace_bytes));

    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    SGX_CALL(sgx_layer_forward(&d, stream));
    HIP_OK(hipStreamSynchronize(stream));

    std::vector<T> out((size_t)N * a.p);
    HIP_OK(hipMemcpy(out.data(), D, out.size() * sizeof(T), hipMemcpyDeviceToHost));
    const int cols = a.cols > 0 && a.cols < a.p ? a.cols : a.p;
    std::stringstream rows(a.rows);
    std::string tok;
    while (std::getline(rows, tok, ',')) {
        const int i = std::atoi(tok.c_str());
        if (i < 0 || i >= N) die("--rows: " + tok + " is not a row of D");
        for (int j = 0; j < cols; ++j)
            std::cout << "out :data index= " << i << " " << j << " kernel = " << (float)out[(size_t)i * a.p + j] << std::endl;
    }

    if (a.time > 0) {
        void *e0, *e1;
        SGX_CALL(sgx_event_create(&e0));
        SGX_CALL(sgx_event_create(&e1));
        SGX_CALL(sgx_event_record(e0, stream));
        for (int it = 0; it < a.time; ++it) SGX_CALL(sgx_layer_forward(&d, stream));
        SGX_CALL(sgx_event_record(e1, stream));
        float ms = 0.0f;
        SGX_CALL(sgx_event_elapsed_ms(e0, e1, &ms));
        std::cout << "layer time: " << ms / a.time << " ms (" << N << " nodes, " << A.col.size() << " edges, P " << a.p
                  << ")" << std::endl;
    }
    return 0;
}

}  // namespace

int main(int argc, char **argv)
{
    const Args a = parse(argc, argv);
    if (a.exact && a.f32) die("--exact reproduces the reference's half arithmetic: f16 only");
    return a.f32 ? run<float>(a) : run<half_t>(a);
}
