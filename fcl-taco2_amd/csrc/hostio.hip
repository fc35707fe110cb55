// hostio.hip — host-side output of the decode driver: Kaldi ark records of a whole batch in ONE call (no device code).
// The reference hands every mel to kaldiio's WriteHelper("ark,scp:...") one utterance at a time (tts.py:652,674); at 36 M frames/s the Python
// side of that (3 writes + a copy per utterance, ~12 us) was the decode driver's bound (20 M frames/s end to end).  Here the records of a batch
// are gathered with writev straight from the pinned landing buffer: one system call per <= 512 utterances, no intermediate copy.
#include <errno.h>
#include <string.h>
#include <sys/uio.h>
#include <unistd.h>

#include <algorithm>
#include <vector>

#include "fcl_common.h"

using namespace fcl;

extern "C" {

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
