// dist_comm.hpp — the collectives of the multi-GPU host (`num_gpus: N`): one PROCESS per GPU, RCCL over xGMI.
//
// The reference has no distributed path; this is the C++ restatement of
// cloud_map_evaluation_amd/dist.py's slab driver for the drop-in binary.  Two transports behind one small interface:
//   RcclComm  production.  ncclUniqueId bootstrap through a file next to the results (rank 0 writes it, the others poll),
//             ncclCommInitRank, collectives on the communicator's own HIP stream.  All buffers are DEVICE memory.
//   FileComm  TESTS ONLY (MAPEVAL_COMM=file): every collective goes through files in a scratch directory, so that several
//             ranks can share ONE GPU (RCCL refuses two ranks on a device) — what tests/test_gpu_host.py uses to run the
//             2-rank code path on the single-GPU box.
#pragma once
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <string>
#include <thread>
#include <vector>

namespace medist {

struct Comm {
    int rank = 0, world = 1;
    std::string err;
    virtual ~Comm() {}
    virtual bool all_reduce_sum_f64(double *dev, size_t n) = 0;
    virtual bool all_reduce_min_f64(double *dev, size_t n) = 0;
    virtual bool all_reduce_sum_u8(uint8_t *dev, size_t n) = 0;
    // recv_dev holds world x bytes, rank k's contribution at offset k * bytes
    virtual bool all_gather(const void *send_dev, void *recv_dev, size_t bytes) = 0;
    // the halo exchange: send_bytes[k] bytes of send_dev (segments back to back, destination-major) go to rank k, recv_bytes[k]
    // bytes arrive from rank k (back to back, source-major).  Device buffers; sizes known on both sides (exchanged beforehand).
    virtual bool all_to_all_v(const void *send_dev, const size_t *send_bytes, void *recv_dev, const size_t *recv_bytes) = 0;
    // true once any rank has called it with ok = false: the ranks agree on "carry on" before a phase that every rank must enter
    // (a rank that failed locally says so here instead of leaving the others blocked in the next collective)
    virtual bool all_ok(bool ok) = 0;
    virtual const char *name() const = 0;
};

inline bool wait_for_file(const std::string &path, size_t min_bytes, double timeout_s) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        std::ifstream f(path, std::ios::binary | std::ios::ate);
        if (f && (size_t) f.tellg() >= min_bytes) return true;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
}
inline bool write_file_atomic(const std::string &path, const void *data, size_t bytes) {
    const std::string tmp = path + ".tmp";
    {
        std::ofstream f(tmp, std::ios::binary);
        if (!f) return false;
        f.write((const char *) data, (std::streamsize) bytes);
        if (!f) return false;
    }
    return std::rename(tmp.c_str(), path.c_str()) == 0;
}

class RcclComm : public Comm {
    ncclComm_t comm_ = nullptr;
    hipStream_t stream_ = nullptr;
    bool ck(ncclResult_t r, const char *what) {
        if (r == ncclSuccess) return true;
        err = std::string(what) + ": " + ncclGetErrorString(r);
        return false;
    }
    bool sync() {
        const hipError_t e = hipStreamSynchronize(stream_);
        if (e == hipSuccess) return true;
        err = std::string("hipStreamSynchronize: ") + hipGetErrorString(e);
        return false;
    }

  public:
    // id_path: where rank 0 publishes the ncclUniqueId (removed by rank 0 in the destructor)
    bool init(int rank_, int world_, int device, const std::string &id_path) {
        rank = rank_;
        world = world_;
        id_path_ = id_path;
        // pre-flight over files, BEFORE the communicator: a rank whose device does not exist (num_gpus larger than the node) would
        // otherwise leave the others blocked inside ncclCommInitRank
        int n_dev = 0;
        const bool dev_ok = hipGetDeviceCount(&n_dev) == hipSuccess && device >= 0 && device < n_dev && hipSetDevice(device) == hipSuccess;
        const char mark = dev_ok ? '1' : '0';
        write_file_atomic(id_path + ".ready" + std::to_string(rank), &mark, 1);
        for (int k = 0; k < world; ++k) {
            const std::string f = id_path + ".ready" + std::to_string(k);
            if (!wait_for_file(f, 1, 120.0)) {
                err = "rank " + std::to_string(k) + " did not start";
                return false;
            }
            char m = '0';
            std::ifstream(f, std::ios::binary).read(&m, 1);
            if (m != '1') {
                err = "rank " + std::to_string(k) + " has no GPU (device " + std::to_string(device - rank + k) + " of " + std::to_string(n_dev) + ")";
                return false;
            }
        }
        ncclUniqueId id;
        if (rank == 0) {
            if (!ck(ncclGetUniqueId(&id), "ncclGetUniqueId")) return false;
            if (!write_file_atomic(id_path, &id, sizeof(id))) {
                err = "cannot write " + id_path;
                return false;
            }
        } else {
            if (!wait_for_file(id_path, sizeof(id), 120.0)) {
                err = "timed out waiting for " + id_path;
                return false;
            }
            std::ifstream f(id_path, std::ios::binary);
            f.read((char *) &id, sizeof(id));
        }
        if (!ck(ncclCommInitRank(&comm_, world, id, rank), "ncclCommInitRank")) return false;
        if (hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking) != hipSuccess) {
            err = "hipStreamCreate failed";
            return false;
        }
        return true;
    }
    ~RcclComm() override {
        const bool was_up = comm_ != nullptr;
        if (comm_) ncclCommDestroy(comm_);
        if (stream_) (void) hipStreamDestroy(stream_);
        // only after a successful init (a barrier: every rank has read the files by then); after a failed one the launcher calls
        // remove_bootstrap_files once its children are gone — a slower rank may still be looking for them
        if (rank == 0 && was_up && !id_path_.empty()) remove_bootstrap_files(id_path_, world);
    }
    static void remove_bootstrap_files(const std::string &id_path, int world) {
        std::remove(id_path.c_str());
        for (int k = 0; k < world; ++k) std::remove((id_path + ".ready" + std::to_string(k)).c_str());
    }
    bool all_reduce_sum_f64(double *dev, size_t n) override {
        return n == 0 || (ck(ncclAllReduce(dev, dev, n, ncclDouble, ncclSum, comm_, stream_), "ncclAllReduce(sum, f64)") && sync());
    }
    bool all_reduce_min_f64(double *dev, size_t n) override {
        return n == 0 || (ck(ncclAllReduce(dev, dev, n, ncclDouble, ncclMin, comm_, stream_), "ncclAllReduce(min, f64)") && sync());
    }
    bool all_reduce_sum_u8(uint8_t *dev, size_t n) override {
        return n == 0 || (ck(ncclAllReduce(dev, dev, n, ncclUint8, ncclSum, comm_, stream_), "ncclAllReduce(sum, u8)") && sync());
    }
    bool all_gather(const void *send_dev, void *recv_dev, size_t bytes) override {
        return bytes == 0 || (ck(ncclAllGather(send_dev, recv_dev, bytes, ncclUint8, comm_, stream_), "ncclAllGather") && sync());
    }
    bool all_to_all_v(const void *send_dev, const size_t *send_bytes, void *recv_dev, const size_t *recv_bytes) override {
        // one group of point-to-point transfers: xGMI is point-to-point, every pair of ranks uses its own link
        if (!ck(ncclGroupStart(), "ncclGroupStart")) return false;
        size_t so = 0, ro = 0;
        bool ok = true;
        for (int k = 0; k < world && ok; ++k) {
            if (send_bytes[k]) ok = ck(ncclSend((const char *) send_dev + so, send_bytes[k], ncclUint8, k, comm_, stream_), "ncclSend");
            if (ok && recv_bytes[k]) ok = ck(ncclRecv((char *) recv_dev + ro, recv_bytes[k], ncclUint8, k, comm_, stream_), "ncclRecv");
            so += send_bytes[k];
            ro += recv_bytes[k];
        }
        const bool ended = ck(ncclGroupEnd(), "ncclGroupEnd");
        return ok && ended && sync();
    }
    bool all_ok(bool ok) override {
        if (!flag_ && hipMalloc((void **) &flag_, 8) != hipSuccess) return false;
        const double v = ok ? 1.0 : 0.0;
        if (hipMemcpy(flag_, &v, 8, hipMemcpyHostToDevice) != hipSuccess) return false;
        if (!all_reduce_min_f64(flag_, 1)) return false;
        double r = 0;
        return hipMemcpy(&r, flag_, 8, hipMemcpyDeviceToHost) == hipSuccess && r > 0.5;
    }
    const char *name() const override { return "rccl"; }

  private:
    std::string id_path_;
    double *flag_ = nullptr;
};

class FileComm : public Comm {  // tests only: see the header comment
    std::string dir_;
    long seq_ = 0;
    std::string path(long s, int r) const { return dir_ + "/c" + std::to_string(s) + "_r" + std::to_string(r); }
    bool exchange(const void *mine, size_t bytes, std::vector<std::vector<char>> &all) {
        const long s = seq_++;
        if (!write_file_atomic(path(s, rank), mine, bytes)) {
            err = "FileComm: cannot write into " + dir_;
            return false;
        }
        all.assign((size_t) world, std::vector<char>(bytes));
        for (int k = 0; k < world; ++k) {
            if (!wait_for_file(path(s, k), bytes, 300.0)) {
                err = "FileComm: timed out waiting for rank " + std::to_string(k);
                return false;
            }
            std::ifstream f(path(s, k), std::ios::binary);
            f.read(all[(size_t) k].data(), (std::streamsize) bytes);
        }
        return true;
    }
    template <typename T, typename F>
    bool reduce(T *dev, size_t n, F op) {
        if (n == 0) return true;
        std::vector<T> h(n);
        if (hipMemcpy(h.data(), dev, n * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) return false;
        std::vector<std::vector<char>> all;
        if (!exchange(h.data(), n * sizeof(T), all)) return false;
        for (size_t i = 0; i < n; ++i) {
            T acc = reinterpret_cast<const T *>(all[0].data())[i];
            for (int k = 1; k < world; ++k) acc = op(acc, reinterpret_cast<const T *>(all[(size_t) k].data())[i]);
            h[i] = acc;
        }
        return hipMemcpy(dev, h.data(), n * sizeof(T), hipMemcpyHostToDevice) == hipSuccess;
    }

  public:
    bool init(int rank_, int world_, const std::string &dir) {
        rank = rank_;
        world = world_;
        dir_ = dir;
        return true;
    }
    bool all_reduce_sum_f64(double *dev, size_t n) override {
        return reduce(dev, n, [](double a, double b) { return a + b; });
    }
    bool all_reduce_min_f64(double *dev, size_t n) override {
        return reduce(dev, n, [](double a, double b) { return b < a ? b : a; });
    }
    bool all_reduce_sum_u8(uint8_t *dev, size_t n) override {
        return reduce(dev, n, [](uint8_t a, uint8_t b) { return (uint8_t) (a + b); });
    }
    bool all_gather(const void *send_dev, void *recv_dev, size_t bytes) override {
        if (bytes == 0) return true;
        std::vector<char> h(bytes);
        if (hipMemcpy(h.data(), send_dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return false;
        std::vector<std::vector<char>> all;
        if (!exchange(h.data(), bytes, all)) return false;
        for (int k = 0; k < world; ++k)
            if (hipMemcpy((char *) recv_dev + (size_t) k * bytes, all[(size_t) k].data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return false;
        return true;
    }
    bool all_to_all_v(const void *send_dev, const size_t *send_bytes, void *recv_dev, const size_t *recv_bytes) override {
        // every rank publishes [world sizes | its whole send buffer]; a receiver cuts its segment out of every file
        size_t total = 0;
        for (int k = 0; k < world; ++k) total += send_bytes[k];
        std::vector<char> h((size_t) world * 8 + total);
        for (int k = 0; k < world; ++k) {
            const unsigned long long b = send_bytes[k];
            std::memcpy(h.data() + (size_t) k * 8, &b, 8);
        }
        if (total && hipMemcpy(h.data() + (size_t) world * 8, send_dev, total, hipMemcpyDeviceToHost) != hipSuccess) return false;
        const long s = seq_++;
        if (!write_file_atomic(path(s, rank), h.data(), h.size())) {
            err = "FileComm: cannot write into " + dir_;
            return false;
        }
        size_t ro = 0;
        for (int k = 0; k < world; ++k) {
            if (!wait_for_file(path(s, k), (size_t) world * 8, 300.0)) {
                err = "FileComm: timed out waiting for rank " + std::to_string(k);
                return false;
            }
            std::ifstream f(path(s, k), std::ios::binary);
            std::vector<unsigned long long> sz((size_t) world);
            f.read((char *) sz.data(), (std::streamsize) world * 8);
            size_t off = (size_t) world * 8;
            for (int j = 0; j < rank; ++j) off += (size_t) sz[(size_t) j];
            if ((size_t) sz[(size_t) rank] != recv_bytes[k]) {
                err = "FileComm: all_to_all_v size mismatch";
                return false;
            }
            if (recv_bytes[k]) {
                std::vector<char> seg(recv_bytes[k]);
                f.seekg((std::streamoff) off);
                f.read(seg.data(), (std::streamsize) recv_bytes[k]);
                if (!f || hipMemcpy((char *) recv_dev + ro, seg.data(), recv_bytes[k], hipMemcpyHostToDevice) != hipSuccess) return false;
            }
            ro += recv_bytes[k];
        }
        return true;
    }
    bool all_ok(bool ok) override {
        double v = ok ? 1.0 : 0.0;
        std::vector<std::vector<char>> all;
        if (!exchange(&v, 8, all)) return false;
        for (int k = 0; k < world; ++k) {
            double r;
            std::memcpy(&r, all[(size_t) k].data(), 8);
            if (!(r > 0.5)) return false;
        }
        return true;
    }
    const char *name() const override { return "file (tests only)"; }
};

// A device buffer that frees itself (the host side owns only what the collectives carry; clouds live inside me_ctx)
struct DevMem {
    void *p = nullptr;
    size_t bytes = 0;
    ~DevMem() {
        if (p) (void) hipFree(p);
    }
    bool ensure(size_t n) {
        if (n <= bytes && p) return true;
        if (p) (void) hipFree(p);
        p = nullptr;
        bytes = 0;
        if (hipMalloc(&p, n ? n : 8) != hipSuccess) return false;
        bytes = n ? n : 8;
        return true;
    }
    template <typename T>
    T *as() {
        return reinterpret_cast<T *>(p);
    }
};

}  // namespace medist
