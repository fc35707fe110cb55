// hostio.hip — the host's two ends of a synthesis pass: the input feed of a capacity graph (a graph node that pulls the packed batch out of pinned
// host memory) and the decode driver's output, Kaldi ark records of a whole batch in ONE call.
// The reference hands every mel to kaldiio's WriteHelper("ark,scp:...") one utterance at a time (tts.py:652,674); at 36 M frames/s the Python
// side of that (3 writes + a copy per utterance, ~12 us) was the decode driver's bound (20 M frames/s end to end).  Here the records of a batch
// are gathered with writev straight from the pinned landing buffer: one system call per <= 512 utterances, no intermediate copy.
#include <errno.h>
#include <string.h>
#include <sys/uio.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <vector>

#include "fcl_common.h"

using namespace fcl;

namespace fcl {

// The input block of a capacity graph (ids, segment bounds, durations, lengths, pad mask, speaker vectors: ~70 KB) comes out of PINNED host memory
// by a kernel that is the graph's first node, instead of a hipMemcpyAsync in front of every launch: the runtime's copy call costs the host ~80 us
// per pass (measured: as much as enqueueing the 90-node graph), the kernel costs nothing beside the graph launch and ~5 us of one workgroup on
// the GPU (the reads cross PCIe once; fine-grained host memory is uncached on the device, so every pass sees what the host packed).  When every
// read has returned the kernel publishes a sequence number in host memory: the host packs the next batch into the same block once it sees the
// number of its last launch.
__global__ __launch_bounds__(1024) void feed_copy_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, int n16, unsigned int* seq_dev,
                                                         unsigned int* seq_host, unsigned int* bump) {
    // system-scope loads: past every cache of the device whatever the mapping of the host block (what a previous pass read must never be served again)
    const unsigned long long* s64 = reinterpret_cast<const unsigned long long*>(src);
    unsigned long long* d64 = reinterpret_cast<unsigned long long*>(dst);
    for (int i = threadIdx.x; i < 2 * n16; i += 1024) d64[i] = __hip_atomic_load(s64 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();  // a thread issues its stores with the loaded data: past this barrier every read of the host block has returned
    if (threadIdx.x == 0) {
        const unsigned int s = *seq_dev + 1;
        *seq_dev = s;
        if (bump) *bump += 1;
        __hip_atomic_store(seq_host, s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- which streams share a compute pipe (round 6) ---------------------------------------------------------------------------------------------------
// On MI355X the HSA queue behind a HIP stream sits on one of FOUR compute pipes (queue index mod 4, in order of queue creation in the process), and a pipe
// advances one of its queues at a time: two chains of dependent launches on streams of the same pipe take 1.43x the time of one chain, on the same HSA queue
// (more streams than GPU_MAX_HW_QUEUES) 2.0x, on different pipes 1.0x (tools/probe/queue_pipe_probe.hip, profiles/r6_queue_pipe_probe.log).  For the KD update
// that is 8.4 ms against 12.7 ms (the frozen teacher's stream on the student's pipe), decided by how many streams the process happened to create before.  HIP
// does not say which pipe a stream is on, so the library MEASURES: two short chains of ~20 us launches, alone and together.
__global__ __launch_bounds__(256) void pipe_probe_kernel(float* p, int iters) {
    float v = p[blockIdx.x * 256 + threadIdx.x];
    for (int i = 0; i < iters; ++i) v = fmaf(v, 1.0000001f, 1e-7f);
    p[blockIdx.x * 256 + threadIdx.x] = v;
}

namespace {
constexpr int kProbeBlocks = 32, kProbeLaunches = 16, kProbeIters = 1200;  // 16 dependent ~20 us launches per chain: ~0.35 ms alone, 0.35 - 0.9 ms for a pair

struct ProbeBuf {
    float* p[2] = {nullptr, nullptr};
    int init() {
        for (float*& q : p) {
            FCL_HIP(hipMalloc(&q, kProbeBlocks * 256 * sizeof(float)));
            FCL_HIP(hipMemset(q, 0, kProbeBlocks * 256 * sizeof(float)));
        }
        return 0;
    }
    ~ProbeBuf() {
        for (float* q : p)
            if (q) (void)hipFree(q);
    }
};

// wall time of kProbeLaunches dependent launches on a (and, at the same time, on b when pair); the two streams are idle before and after
int chain_ms(hipStream_t a, hipStream_t b, bool pair, const ProbeBuf& buf, double* ms) {
    double best = 1e30;
    for (int rep = 0; rep < 2; ++rep) {
        FCL_HIP(hipStreamSynchronize(a));
        if (pair) FCL_HIP(hipStreamSynchronize(b));
        const auto t0 = std::chrono::steady_clock::now();
        for (int l = 0; l < kProbeLaunches; ++l) {
            hipLaunchKernelGGL(pipe_probe_kernel, dim3(kProbeBlocks), dim3(256), 0, a, buf.p[0], kProbeIters);
            if (pair) hipLaunchKernelGGL(pipe_probe_kernel, dim3(kProbeBlocks), dim3(256), 0, b, buf.p[1], kProbeIters);
        }
        FCL_HIP(hipStreamSynchronize(a));
        if (pair) FCL_HIP(hipStreamSynchronize(b));
        best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    FCL_HIP(hipGetLastError());
    *ms = best;
    return 0;
}

// alone_ms > 0: the single-chain time measured before (one measurement serves every pair a candidate stream is tested in)
int share_pipe(hipStream_t a, hipStream_t b, const ProbeBuf& buf, bool* shared, double* ratio, double alone_ms = 0.0) {
    if (a == b) {
        *shared = true;
        if (ratio) *ratio = 2.0;
        return 0;
    }
    double alone = alone_ms, both = 0.0;
    int rc = alone > 0.0 ? 0 : chain_ms(a, b, false, buf, &alone);
    if (rc) return rc;
    rc = chain_ms(a, b, true, buf, &both);
    if (rc) return rc;
    const double r = both / std::max(alone, 1e-6);
    *shared = r > 1.2;  // measured: 1.00 - 1.01 apart, 1.42 - 1.44 same pipe, 2.00 same queue
    if (ratio) *ratio = r;
    return 0;
}
}  // namespace

}  // namespace fcl

extern "C" {

/* Do two streams' queues sit on the same compute pipe?  Measured (two chains of 16 dependent ~20 us launches, alone and together: ~2 ms); both streams must be
 * idle.  *ratio (optional) = pair time / alone: ~1.0 apart, ~1.43 same pipe, ~2.0 same hardware queue. */
int fcl_streams_share_pipe(fcl_stream_t a, fcl_stream_t b, int* shared, double* ratio) {
    FCL_REQUIRE(shared, FCL_ERR_INVALID, "streams_share_pipe: null argument");
    ProbeBuf buf;
    int rc = buf.init();
    if (rc) return rc;
    bool sh = false;
    rc = share_pipe((hipStream_t)a, (hipStream_t)b, buf, &sh, ratio);
    if (rc) return rc;
    *shared = sh ? 1 : 0;
    return 0;
}

/* A new stream whose queue shares a pipe with none of others[0 .. n): candidates are created until one measures apart from all of them (at most 24; the
 * rejected ones are destroyed afterwards, not before -- a destroyed stream's queue slot would be handed out again).  n <= 3 can always be satisfied (four pipes);
 * FCL_ERR_HIP when no candidate fits.  *tried (optional) = candidates created. */
int fcl_stream_create_apart(const fcl_stream_t* others, int n, fcl_stream_t* out, int* tried) {
    FCL_REQUIRE(out && n >= 0 && (others || n == 0), FCL_ERR_INVALID, "stream_create_apart: bad arguments");
    ProbeBuf buf;
    int rc = buf.init();
    if (rc) return rc;
    std::vector<hipStream_t> rejected;
    hipStream_t good = nullptr;
    int made = 0;
    for (; made < 24 && !good; ++made) {
        hipStream_t s = nullptr;
        FCL_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        bool clash = false;
        double alone = 0.0;
        if (n > 0) rc = chain_ms(s, s, false, buf, &alone);  // (also the new queue's warm-up)
        for (int k = 0; k < n && !clash && !rc; ++k) rc = share_pipe((hipStream_t)others[k], s, buf, &clash, nullptr, alone);
        if (rc) {
            rejected.push_back(s);
            break;
        }
        if (clash) rejected.push_back(s);
        else good = s;
    }
    for (hipStream_t s : rejected) (void)hipStreamDestroy(s);
    if (tried) *tried = made;
    if (rc) return rc;
    FCL_REQUIRE(good, FCL_ERR_HIP, "stream_create_apart: no stream apart from the %d given ones in %d candidates", n, made);
    *out = (fcl_stream_t)good;
    return 0;
}

int fcl_stream_create_cus(int n_cus, fcl_stream_t* out) {
    FCL_REQUIRE(out, FCL_ERR_INVALID, "stream_create_cus: null argument");
    int dev = 0, total = 0;
    FCL_HIP(hipGetDevice(&dev));
    FCL_HIP(hipDeviceGetAttribute(&total, hipDeviceAttributeMultiprocessorCount, dev));
    hipStream_t s = nullptr;
    if (n_cus <= 0 || n_cus >= total) {
        FCL_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    } else {
        // bit i of the queue's CU mask lands on XCD i % 8 (the driver deals the bits round-robin over the XCCs): the low n_cus bits = n_cus / 8 CUs of every XCD
        uint32_t mask[16] = {};
        for (int i = 0; i < n_cus && i < 512; ++i) mask[i >> 5] |= 1u << (i & 31);
        FCL_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)((total + 31) / 32), mask));
    }
    *out = (fcl_stream_t)s;
    return 0;
}

int fcl_stream_destroy(fcl_stream_t stream) {
    if (stream) FCL_HIP(hipStreamDestroy((hipStream_t)stream));
    return 0;
}

void* fcl_host_device_ptr(void* pinned_host) {
    void* p = nullptr;
    if (!pinned_host || hipHostGetDevicePointer(&p, pinned_host, 0) != hipSuccess) {
        (void)hipGetLastError();
        set_error("host_device_ptr: not pinned, mapped host memory");
        return nullptr;
    }
    return p;
}

int fcl_feed_copy(void* dst, const void* src, size_t bytes, uint32_t* seq_dev, uint32_t* seq_host, uint32_t* bump, fcl_stream_t stream) {
    FCL_REQUIRE(dst && src && seq_dev && seq_host && bytes > 0 && (bytes & 15) == 0 && bytes <= (1u << 26), FCL_ERR_INVALID, "feed_copy: bad arguments");
    FCL_REQUIRE(((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15u) == 0, FCL_ERR_ALIGN, "feed_copy: 16-byte alignment required");
    hipLaunchKernelGGL(feed_copy_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, reinterpret_cast<uint4*>(dst), reinterpret_cast<const uint4*>(src),
                       (int)(bytes / 16), seq_dev, seq_host, bump);
    return check_hip(hipGetLastError(), "feed_copy");
}

// Record layout (Kaldi binary FloatMatrix): <key> ' ' '\0' 'B' 'F' 'M' ' ' '\4' <int32 rows> '\4' <int32 cols> <rows*cols float32>.
// offsets[i] (scp): byte offset of utterance i's '\0B' marker in the file, given that the first byte written lands at file_pos.
long long fcl_kaldi_ark_append(int fd, long long file_pos, int n, const char* const* keys, const float* data, const int* rows, int cols,
                               long long* offsets) {
    if (fd < 0 || n < 0 || cols <= 0 || (n > 0 && (!keys || !data || !rows || !offsets))) {
        set_error("kaldi_ark_append: bad arguments");
        return FCL_ERR_INVALID;
    }
    struct Hdr { char b[15]; };  // "\0BFM " '\4' rows '\4' cols
    std::vector<Hdr> hdr((size_t)n);
    std::vector<struct iovec> iov;
    iov.reserve((size_t)n * 4);
    static const char space = ' ';
    long long pos = file_pos;
    const float* p = data;
    for (int i = 0; i < n; ++i) {
        if (!keys[i] || rows[i] < 0 || strchr(keys[i], ' ')) {
            set_error("kaldi_ark_append: bad key / row count at entry %d", i);
            return FCL_ERR_INVALID;
        }
        const size_t kl = strlen(keys[i]);
        char* h = hdr[(size_t)i].b;
        memcpy(h, "\0BFM \4", 6);
        const int32_t r = rows[i], c = cols;
        memcpy(h + 6, &r, 4);
        h[10] = '\4';
        memcpy(h + 11, &c, 4);
        iov.push_back({const_cast<char*>(keys[i]), kl});
        iov.push_back({const_cast<char*>(&space), 1});
        iov.push_back({h, 15});
        const size_t bytes = sizeof(float) * (size_t)r * (size_t)c;
        if (bytes) iov.push_back({const_cast<float*>(p), bytes});
        offsets[i] = pos + (long long)kl + 1;
        pos += (long long)kl + 1 + 15 + (long long)bytes;
        p += (size_t)r * (size_t)c;
    }
    size_t at = 0;
    while (at < iov.size()) {  // writev takes <= IOV_MAX entries and may write short
        const int cnt = (int)std::min<size_t>(iov.size() - at, 512);
        ssize_t w = writev(fd, iov.data() + at, cnt);
        if (w < 0) {
            if (errno == EINTR) continue;
            set_error("kaldi_ark_append: writev failed: %s", strerror(errno));
            return FCL_ERR_INVALID;
        }
        size_t left = (size_t)w;
        while (left > 0 && at < iov.size()) {
            if (left >= iov[at].iov_len) {
                left -= iov[at].iov_len;
                ++at;
            } else {
                iov[at].iov_base = static_cast<char*>(iov[at].iov_base) + left;
                iov[at].iov_len -= left;
                left = 0;
            }
        }
        while (at < iov.size() && iov[at].iov_len == 0) ++at;
    }
    return pos;
}

}  // extern "C"
