// Pinned-host -> HBM asynchronous staging ring + .npy header parser.
//
// Replaces the reference's synchronous numpy_reader (src/numpy_reader/cpp/
// numpy_reader.{h,cpp}: parse :60-170, chunked pread/cuFileRead :403-431,
// per-sample blocking RunAll :458-475) on the data path of the training step:
//
//   submit(path, offset, bytes)  -> ticket      (returns at once)
//       worker threads pread() disjoint chunks of the payload into a pinned
//       host slot; the thread that finishes the last chunk enqueues the
//       H2D copy of the slot on the ring's own HIP stream and records an event
//   acquire(ticket, consumer_stream) -> device pointer
//       the consumer stream waits on that event (no host block once the file
//       read has completed); with device = -1 the host pointer is returned
//   release(ticket)               -> slot may be reused
//
// so file reads and PCIe copies of sample k+1.. overlap the step on sample k.
// A short read is an error ("file corruption"), not an endless loop
// (numpy_reader.cpp:412-430 spins when pread returns 0).
#include "common.h"

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <errno.h>
#include <fcntl.h>
#include <functional>
#include <mutex>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

#define BG_E_IO (-3)

namespace {

struct Slot {
    char* host = nullptr;
    char* dev = nullptr;
    hipEvent_t ev = nullptr;
    int64_t ticket = -1;
    int64_t nbytes = 0;
    int pending = 0;     // chunks still being read
    int state = 0;       // 0 free, 1 reading, 2 ready, 3 failed
    std::string error;
};

}  // namespace

struct bg_ring {
    int device = -1;
    int64_t slot_bytes = 0;
    std::vector<Slot> slots;
    hipStream_t copy_stream = nullptr;
    std::vector<std::thread> workers;
    std::deque<std::function<void()>> jobs;
    std::mutex mu;
    std::condition_variable cv_jobs, cv_slots;
    bool stop = false;
    int64_t next_ticket = 0;
};

namespace {

void worker_main(bg_ring* r) {
    if (r->device >= 0) (void)hipSetDevice(r->device);
    for (;;) {
        std::function<void()> job;
        {
            std::unique_lock<std::mutex> lk(r->mu);
            r->cv_jobs.wait(lk, [&] { return r->stop || !r->jobs.empty(); });
            if (r->stop && r->jobs.empty()) return;
            job = std::move(r->jobs.front());
            r->jobs.pop_front();
        }
        job();
    }
}

// Read [off, off+n) of fd into dst; returns "" or an error message.
std::string read_fully(int fd, char* dst, int64_t off, int64_t n) {
    int64_t done = 0;
    while (done < n) {
        const ssize_t got = pread(fd, dst + done, (size_t)(n - done), (off_t)(off + done));
        if (got < 0) {
            if (errno == EINTR) continue;
            return std::string("read error: ") + strerror(errno);
        }
        if (got == 0) return "file corruption: unexpected end of file (payload shorter than the header promises)";
        done += got;
    }
    return "";
}

Slot* find_slot(bg_ring* r, int64_t ticket) {
    for (auto& s : r->slots)
        if (s.ticket == ticket && s.state != 0) return &s;
    return nullptr;
}

}  // namespace

// ------------------------------------------------------------------- npy ----
extern "C" int bg_npy_parse(const char* path, bg_npy_info* out) {
    BG_CHECK_ARG(path && out, "bg_npy_parse: null argument");
    memset(out, 0, sizeof(*out));
    const int fd = open(path, O_RDONLY);
    if (fd < 0) {
        bg_set_error("bg_npy_parse: failed to open file %s: %s", path, strerror(errno));
        return BG_E_IO;
    }
    struct stat st;
    fstat(fd, &st);
    out->file_size = (int64_t)st.st_size;
    unsigned char head[12];
    const ssize_t got = pread(fd, head, 12, 0);
    auto fail = [&](const char* msg) {
        close(fd);
        bg_set_error("bg_npy_parse: %s (%s)", msg, path);
        return BG_E_IO;
    };
    if (got < 10) return fail("not a numpy file");
    if (memcmp(head, "\x93NUMPY", 6) != 0) {
        // a header that lost its magic/version bytes still starts like the dict
        if (memchr(head, '{', 10) != nullptr) return fail("cannot parse header, ill formatted or corrupt");
        return fail("not a numpy file");
    }
    const int major = head[6];
    int64_t hlen, hoff;
    if (major == 1) {
        hlen = head[8] | (head[9] << 8);
        hoff = 10;
    } else if (major == 2 || major == 3) {
        if (got < 12) return fail("cannot parse header, ill formatted or corrupt");
        hlen = (int64_t)head[8] | ((int64_t)head[9] << 8) | ((int64_t)head[10] << 16) | ((int64_t)head[11] << 24);
        hoff = 12;
    } else {
        return fail("cannot parse header, ill formatted or corrupt");
    }
    if (hlen <= 0 || hlen > (1 << 20) || hoff + hlen > out->file_size) return fail("cannot parse header, ill formatted or corrupt");
    std::string h((size_t)hlen, '\0');
    if (pread(fd, &h[0], (size_t)hlen, (off_t)hoff) != hlen) return fail("cannot parse header, ill formatted or corrupt");
    close(fd);
    out->data_offset = hoff + hlen;

    auto value_of = [&](const char* key) -> std::string {
        const size_t k = h.find(key);
        if (k == std::string::npos) return "";
        size_t c = h.find(':', k);
        if (c == std::string::npos) return "";
        ++c;
        while (c < h.size() && h[c] == ' ') ++c;
        return h.substr(c);
    };
    std::string descr = value_of("'descr'");
    std::string order = value_of("'fortran_order'");
    std::string shape = value_of("'shape'");
    if (descr.empty() || order.empty() || shape.empty() || descr[0] != '\'' || shape[0] != '(') {
        bg_set_error("bg_npy_parse: cannot parse header, ill formatted or corrupt (%s)", path);
        return BG_E_IO;
    }
    const size_t q = descr.find('\'', 1);
    if (q == std::string::npos || q < 3) {
        bg_set_error("bg_npy_parse: cannot parse header, ill formatted or corrupt (%s)", path);
        return BG_E_IO;
    }
    const std::string ts = descr.substr(1, q - 1);  // e.g. "<f4"
    if (ts[0] == '>') {
        bg_set_error("bg_npy_parse: the specified file is in big endian. This is currently not supported. (%s)", path);
        return BG_E_IO;
    }
    const std::string tid = ts.substr(1);
    if (tid == "f4") { out->dtype_code = 0; out->typesize = 4; }
    else if (tid == "f8") { out->dtype_code = 1; out->typesize = 8; }
    else if (tid == "i4") { out->dtype_code = 2; out->typesize = 4; }
    else if (tid == "i8") { out->dtype_code = 3; out->typesize = 8; }
    else {
        bg_set_error("bg_npy_parse: unsupported datatype %s (%s)", ts.c_str(), path);
        return BG_E_IO;
    }
    out->fortran_order = order.compare(0, 4, "True") == 0 ? 1 : 0;
    // shape: "(a, b, c), ..." ; "()" = scalar
    const size_t close_paren = shape.find(')');
    if (close_paren == std::string::npos) {
        bg_set_error("bg_npy_parse: cannot parse header, ill formatted or corrupt (%s)", path);
        return BG_E_IO;
    }
    const std::string dims = shape.substr(1, close_paren - 1);
    out->ndim = 0;
    size_t pos = 0;
    while (pos < dims.size()) {
        while (pos < dims.size() && (dims[pos] == ' ' || dims[pos] == ',')) ++pos;
        if (pos >= dims.size()) break;
        char* end = nullptr;
        const long long v = strtoll(dims.c_str() + pos, &end, 10);
        if (end == dims.c_str() + pos || v < 0 || out->ndim >= 8) {
            bg_set_error("bg_npy_parse: cannot parse header, ill formatted or corrupt (%s)", path);
            return BG_E_IO;
        }
        out->shape[out->ndim++] = v;
        pos = (size_t)(end - dims.c_str());
    }
    return BG_OK;
}

// ------------------------------------------------------------------ ring ----
extern "C" int bg_ring_create(int32_t device, int32_t n_slots, int64_t slot_bytes, int32_t n_threads, bg_ring** out) {
    BG_CHECK_ARG(out && n_slots >= 1 && n_slots <= 64 && slot_bytes > 0 && n_threads >= 1 && n_threads <= 64,
                 "bg_ring_create: bad arguments");
    bg_ring* r = new bg_ring();
    r->device = device;
    r->slot_bytes = slot_bytes;
    r->slots.resize(n_slots);
    if (device >= 0) {
        if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&r->copy_stream, hipStreamNonBlocking) != hipSuccess) {
            delete r;
            bg_set_error("bg_ring_create: cannot use device %d", device);
            return BG_E_LAUNCH;
        }
    }
    for (auto& s : r->slots) {
        bool ok = true;
        if (device >= 0) {
            ok = hipHostMalloc((void**)&s.host, (size_t)slot_bytes, hipHostMallocDefault) == hipSuccess &&
                 hipMalloc((void**)&s.dev, (size_t)slot_bytes) == hipSuccess &&
                 hipEventCreateWithFlags(&s.ev, hipEventDisableTiming) == hipSuccess;
        } else {
            ok = posix_memalign((void**)&s.host, 4096, (size_t)slot_bytes) == 0;
        }
        if (!ok) {
            bg_set_error("bg_ring_create: allocation of %lld-byte slots failed", (long long)slot_bytes);
            *out = r;
            bg_ring_destroy(r);
            *out = nullptr;
            return BG_E_LAUNCH;
        }
    }
    for (int i = 0; i < n_threads; ++i) r->workers.emplace_back(worker_main, r);
    *out = r;
    return BG_OK;
}

extern "C" int bg_ring_destroy(bg_ring* r) {
    if (!r) return BG_OK;
    {
        std::lock_guard<std::mutex> lk(r->mu);
        r->stop = true;
    }
    r->cv_jobs.notify_all();
    for (auto& t : r->workers) t.join();
    if (r->device >= 0) {
        (void)hipSetDevice(r->device);
        if (r->copy_stream) (void)hipStreamSynchronize(r->copy_stream);
    }
    for (auto& s : r->slots) {
        if (r->device >= 0) {
            if (s.ev) (void)hipEventDestroy(s.ev);
            if (s.dev) (void)hipFree(s.dev);
            if (s.host) (void)hipHostFree(s.host);
        } else {
            free(s.host);
        }
    }
    if (r->copy_stream) (void)hipStreamDestroy(r->copy_stream);
    delete r;
    return BG_OK;
}

extern "C" int bg_ring_submit(bg_ring* r, const char* path, int64_t offset, int64_t nbytes, int32_t n_chunks,
                              int64_t* ticket) {
    BG_CHECK_ARG(r && path && ticket && offset >= 0 && nbytes > 0, "bg_ring_submit: bad arguments");
    BG_CHECK_ARG(nbytes <= r->slot_bytes, "bg_ring_submit: %lld bytes do not fit a %lld-byte slot", (long long)nbytes,
                 (long long)r->slot_bytes);
    const int fd = open(path, O_RDONLY);
    if (fd < 0) {
        bg_set_error("bg_ring_submit: failed to open file %s: %s", path, strerror(errno));
        return BG_E_IO;
    }
    Slot* s = nullptr;
    {
        std::unique_lock<std::mutex> lk(r->mu);
        for (auto& c : r->slots)
            if (c.state == 0) { s = &c; break; }
        if (!s) {
            close(fd);
            bg_set_error("bg_ring_submit: all %d slots are in flight; release one first", (int)r->slots.size());
            return BG_E_ARG;
        }
        if (n_chunks < 1) n_chunks = 1;
        if ((int64_t)n_chunks > nbytes) n_chunks = 1;
        s->state = 1;
        s->ticket = r->next_ticket++;
        s->nbytes = nbytes;
        s->pending = n_chunks;
        s->error.clear();
        *ticket = s->ticket;
        const int64_t chunk = (nbytes + n_chunks - 1) / n_chunks;
        auto fd_refs = std::make_shared<std::atomic<int>>(n_chunks);
        for (int i = 0; i < n_chunks; ++i) {
            const int64_t c0 = (int64_t)i * chunk;
            const int64_t cn = std::min(chunk, nbytes - c0);
            r->jobs.emplace_back([r, s, fd, offset, c0, cn, fd_refs] {
                std::string err = cn > 0 ? read_fully(fd, s->host + c0, offset + c0, cn) : std::string();
                if (fd_refs->fetch_sub(1) == 1) close(fd);
                std::unique_lock<std::mutex> lk2(r->mu);
                if (!err.empty() && s->error.empty()) s->error = err;
                if (--s->pending == 0) {
                    if (!s->error.empty()) {
                        s->state = 3;
                    } else {
                        if (r->device >= 0) {
                            // the thread that lands the last chunk hands the slot to the copy engine
                            hipError_t e = hipMemcpyAsync(s->dev, s->host, (size_t)s->nbytes, hipMemcpyHostToDevice, r->copy_stream);
                            if (e == hipSuccess) e = hipEventRecord(s->ev, r->copy_stream);
                            if (e != hipSuccess) {
                                s->error = std::string("H2D copy failed: ") + hipGetErrorString(e);
                                s->state = 3;
                            } else {
                                s->state = 2;
                            }
                        } else {
                            s->state = 2;
                        }
                    }
                    r->cv_slots.notify_all();
                }
            });
        }
    }
    r->cv_jobs.notify_all();
    return BG_OK;
}

extern "C" int bg_ring_acquire(bg_ring* r, int64_t ticket, void* consumer_stream, void** ptr) {
    BG_CHECK_ARG(r && ptr, "bg_ring_acquire: bad arguments");
    std::unique_lock<std::mutex> lk(r->mu);
    Slot* s = find_slot(r, ticket);
    BG_CHECK_ARG(s != nullptr, "bg_ring_acquire: unknown ticket %lld", (long long)ticket);
    r->cv_slots.wait(lk, [&] { return s->state >= 2; });
    if (s->state == 3) {
        bg_set_error("bg_ring_acquire: %s", s->error.c_str());
        return BG_E_IO;
    }
    if (r->device >= 0) {
        const hipError_t e = hipStreamWaitEvent((hipStream_t)consumer_stream, s->ev, 0);
        if (e != hipSuccess) {
            bg_set_error("bg_ring_acquire: hipStreamWaitEvent: %s", hipGetErrorString(e));
            return BG_E_LAUNCH;
        }
        *ptr = s->dev;
    } else {
        *ptr = s->host;
    }
    return BG_OK;
}

// Copy the slot's payload to `dst` (device memory on consumer_stream for a device ring,
// host memory otherwise); the usual way to hand a sample to the framework's allocator.
extern "C" int bg_ring_copy_out(bg_ring* r, int64_t ticket, void* dst, int64_t nbytes, void* consumer_stream) {
    void* src = nullptr;
    int rc = bg_ring_acquire(r, ticket, consumer_stream, &src);
    if (rc) return rc;
    if (r->device >= 0) {
        const hipError_t e = hipMemcpyAsync(dst, src, (size_t)nbytes, hipMemcpyDeviceToDevice, (hipStream_t)consumer_stream);
        if (e != hipSuccess) {
            bg_set_error("bg_ring_copy_out: %s", hipGetErrorString(e));
            return BG_E_LAUNCH;
        }
    } else {
        memcpy(dst, src, (size_t)nbytes);
    }
    return BG_OK;
}

extern "C" int bg_ring_release(bg_ring* r, int64_t ticket, void* consumer_stream) {
    BG_CHECK_ARG(r, "bg_ring_release: null ring");
    // the consumer may still be reading the device slot: the next H2D into it must come after
    if (r->device >= 0) {  // (a null handle is the default stream, also valid)
        hipEvent_t done;
        if (hipEventCreateWithFlags(&done, hipEventDisableTiming) == hipSuccess) {
            (void)hipEventRecord(done, (hipStream_t)consumer_stream);
            (void)hipStreamWaitEvent(r->copy_stream, done, 0);
            (void)hipEventDestroy(done);
        }
    }
    std::unique_lock<std::mutex> lk(r->mu);
    Slot* s = find_slot(r, ticket);
    BG_CHECK_ARG(s != nullptr, "bg_ring_release: unknown ticket %lld", (long long)ticket);
    r->cv_slots.wait(lk, [&] { return s->state >= 2; });  // never free a slot that workers still write
    if (r->device >= 0 && s->state == 2) {
        // The H2D copy of this ticket reads the PINNED HOST half of the slot asynchronously.  The stream waits above
        // order the device half only; the next submit's worker threads pread() into s->host as soon as the slot is
        // free, so the host must see that copy finished before the slot changes hands (the slot stays 'in use'
        // meanwhile, nobody else touches it; the wait is outside the lock so the workers of other slots keep going).
        hipEvent_t ev = s->ev;
        lk.unlock();
        const hipError_t e = hipEventSynchronize(ev);
        lk.lock();
        if (e != hipSuccess) {
            bg_set_error("bg_ring_release: waiting for the H2D copy of ticket %lld: %s", (long long)ticket, hipGetErrorString(e));
            return BG_E_LAUNCH;
        }
    }
    s->state = 0;
    s->ticket = -1;
    return BG_OK;
}
