// Host-only helper of the cooler reader (hicpeaks_amd/cool.py): the chunks of a pixel-table column as HDF5 stores them - deflate,
// optionally behind the byte-shuffle filter - inflated, un-shuffled and widened to int64 / f64 (small integers: int32) on a pool of threads with scratch
// buffers of their own.  The reference reads the same columns through cooler / h5py (scripts/pyHICCUPS:142-143), whose filter
// pipeline runs one chunk at a time under HDF5's global lock; the Python-level pool this replaces (zlib.decompress + numpy
// transposes) gave every chunk three fresh megabyte-sized allocations - mmap / munmap under the process' one address-space lock,
// which is what stopped it from scaling past ~16 threads (profiles/r05_host_e2e_deep.txt).
#include <dlfcn.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#include "../../include/hpk.h"

namespace {

template <class T, class O>
void widen(const unsigned char* p, int64_t lo, int64_t hi, O* out, O bias) {
    const T* v = reinterpret_cast<const T*>(p);
    for (int64_t i = lo; i < hi; ++i) out[i - lo] = (O)v[i] - bias;
}

// ... straight out of the shuffled planes (HDF5's shuffle filter: byte b of element e sits at plane b, position e): the element is
// put together from its planes and widened in one pass, no un-shuffled copy of the chunk in between
template <class T, class O>
void unshuffle_widen(const unsigned char* planes, int64_t ne, int64_t lo, int64_t hi, O* out, O bias) {
    for (int64_t e = lo; e < hi; ++e) {
        T v;
        unsigned char* vb = reinterpret_cast<unsigned char*>(&v);
        for (size_t b = 0; b < sizeof(T); ++b) vb[b] = planes[b * (size_t)ne + (size_t)e];
        out[e - lo] = (O)v - bias;
    }
}
template <class O>
bool unshuffle_widen_any(const unsigned char* p, int64_t ne, int32_t size, int32_t kind, int64_t lo, int64_t hi, O* out, O bias) {
    if (kind == 2) {
        if (size == 4) unshuffle_widen<float, O>(p, ne, lo, hi, out, bias); else if (size == 8) unshuffle_widen<double, O>(p, ne, lo, hi, out, bias); else return false;
    } else if (kind == 1) {
        if (size == 2) unshuffle_widen<uint16_t, O>(p, ne, lo, hi, out, bias); else if (size == 4) unshuffle_widen<uint32_t, O>(p, ne, lo, hi, out, bias);
        else if (size == 8) unshuffle_widen<uint64_t, O>(p, ne, lo, hi, out, bias); else return false;
    } else {
        if (size == 2) unshuffle_widen<int16_t, O>(p, ne, lo, hi, out, bias); else if (size == 4) unshuffle_widen<int32_t, O>(p, ne, lo, hi, out, bias);
        else if (size == 8) unshuffle_widen<int64_t, O>(p, ne, lo, hi, out, bias); else return false;
    }
    return true;
}

template <class O>
bool widen_any(const unsigned char* p, int32_t size, int32_t kind, int64_t lo, int64_t hi, O* out, O bias) {
    if (kind == 2) {
        if (size == 4) widen<float, O>(p, lo, hi, out, bias); else if (size == 8) widen<double, O>(p, lo, hi, out, bias); else return false;
    } else if (kind == 1) {
        if (size == 1) widen<uint8_t, O>(p, lo, hi, out, bias); else if (size == 2) widen<uint16_t, O>(p, lo, hi, out, bias);
        else if (size == 4) widen<uint32_t, O>(p, lo, hi, out, bias); else if (size == 8) widen<uint64_t, O>(p, lo, hi, out, bias); else return false;
    } else {
        if (size == 1) widen<int8_t, O>(p, lo, hi, out, bias); else if (size == 2) widen<int16_t, O>(p, lo, hi, out, bias);
        else if (size == 4) widen<int32_t, O>(p, lo, hi, out, bias); else if (size == 8) widen<int64_t, O>(p, lo, hi, out, bias); else return false;
    }
    return true;
}

// libdeflate, if the host has it (loaded at run time: the image ships the library without its header), inflates a chunk about
// twice as fast as zlib; the same zlib-wrapped streams, checked by its own Adler-32.  HPK_NO_LIBDEFLATE=1: zlib.
// zlib is loaded at run time as well (`uncompress`): a build host without its development package still builds the library - GPU
// path included - and a host without either inflater gets HPK_ERR_INVALID here, which sends the reader down its Python path.
struct Deflate {
    void* (*alloc)() = nullptr;
    int (*zlib_decompress)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    void (*free_)(void*) = nullptr;
    int (*z_uncompress)(unsigned char*, unsigned long*, const unsigned char*, unsigned long) = nullptr;      // zlib's uncompress: 0 = Z_OK
    Deflate() {
        {
            void* z = dlopen("libz.so.1", RTLD_NOW | RTLD_LOCAL);
            if (!z) z = dlopen("libz.so", RTLD_NOW | RTLD_LOCAL);
            if (z) z_uncompress = reinterpret_cast<int (*)(unsigned char*, unsigned long*, const unsigned char*, unsigned long)>(dlsym(z, "uncompress"));
        }
        if (std::getenv("HPK_NO_LIBDEFLATE")) return;
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = reinterpret_cast<void* (*)()>(dlsym(h, "libdeflate_alloc_decompressor"));
        zlib_decompress = reinterpret_cast<int (*)(void*, const void*, size_t, void*, size_t, size_t*)>(dlsym(h, "libdeflate_zlib_decompress"));
        free_ = reinterpret_cast<void (*)(void*)>(dlsym(h, "libdeflate_free_decompressor"));
        if (!alloc || !zlib_decompress || !free_) alloc = nullptr;
    }
};
const Deflate& deflate_lib() { static const Deflate d; return d; }

}  // namespace

namespace {
int decode_impl(const void* const* src, int fd, const uint64_t* file_off, const uint64_t* src_len, int64_t nchunks, int64_t first_chunk,
                int64_t chunk_elems, int32_t elem_size, int32_t kind, int32_t shuffle, int64_t start, int64_t stop, void* out,
                int32_t out_f64, int64_t bias, int32_t threads) {
    if ((!src && (fd < 0 || !file_off)) || !src_len || !out || nchunks < 0 || chunk_elems <= 0 || stop < start || kind < 0 || kind > 2) return HPK_ERR_INVALID;
    if (elem_size != 1 && elem_size != 2 && elem_size != 4 && elem_size != 8) return HPK_ERR_INVALID;
    // (int32 output: integer columns that fit - signed up to four bytes, unsigned up to two)
    if (out_f64 < 0 || out_f64 > 2 || (out_f64 == 2 && !((kind == 0 && elem_size <= 4) || (kind == 1 && elem_size <= 2)))) return HPK_ERR_INVALID;
    const size_t cbytes = (size_t)chunk_elems * (size_t)elem_size;
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(threads > 0 ? threads : (int)std::thread::hardware_concurrency(), nchunks));
    std::atomic<int64_t> next{0};
    std::atomic<int> bad{0};
    auto work = [&]() {
        std::vector<unsigned char> plain(cbytes), stored;
        const Deflate& dl = deflate_lib();
        void* dec = dl.alloc ? dl.alloc() : nullptr;
        struct Guard { const Deflate& d; void* p; ~Guard() { if (p) d.free_(p); } } guard{dl, dec};
        for (;;) {
            const int64_t i = next.fetch_add(1);
            if (i >= nchunks || bad.load()) break;
            const unsigned char* in = src ? static_cast<const unsigned char*>(src[i]) : nullptr;
            if (!src) {             // the chunk as stored, read by this thread (pread: no file position shared)
                stored.resize((size_t)src_len[i]);
                size_t have = 0;
                while (have < stored.size()) {
                    const ssize_t r = pread(fd, stored.data() + have, stored.size() - have, (off_t)(file_off[i] + have));
                    if (r <= 0) break;
                    have += (size_t)r;
                }
                if (have != stored.size()) { bad.store(1); break; }
                in = stored.data();
            }
            unsigned long got = (unsigned long)cbytes;
            bool inflated = false;
            if (dec) {
                size_t actual = 0;
                inflated = dl.zlib_decompress(dec, in, (size_t)src_len[i], plain.data(), cbytes, &actual) == 0;
                got = (unsigned long)actual;
            }
            if (!inflated && dl.z_uncompress) {
                got = (unsigned long)cbytes;
                inflated = dl.z_uncompress(plain.data(), &got, in, (unsigned long)src_len[i]) == 0;
            }
            // (HDF5 stores every chunk whole, the file's last one too: a chunk that inflates to anything else is corrupt - refused, so
            // that the caller falls back to H5Dread instead of keeping what an earlier chromosome left in a recycled array)
            if (!inflated || got != (unsigned long)cbytes) {
                bad.store(1);
                break;
            }
            const int64_t ne = (int64_t)(got / (size_t)elem_size);          // (the file's last chunk is stored whole, too)
            const int64_t c0 = (first_chunk + i) * chunk_elems;
            const int64_t lo = std::max(start, c0), hi = std::min(stop, c0 + ne);
            if (hi <= lo) continue;
            const unsigned char* p = plain.data();
            const bool sh = shuffle && elem_size > 1;
            const int64_t a = lo - c0, b = hi - c0, at = lo - start;
            bool ok;
            if (out_f64 == 2) ok = sh ? unshuffle_widen_any<int32_t>(p, ne, elem_size, kind, a, b, static_cast<int32_t*>(out) + at, (int32_t)bias)
                                      : widen_any<int32_t>(p, elem_size, kind, a, b, static_cast<int32_t*>(out) + at, (int32_t)bias);
            else if (out_f64) ok = sh ? unshuffle_widen_any<double>(p, ne, elem_size, kind, a, b, static_cast<double*>(out) + at, (double)bias)
                                      : widen_any<double>(p, elem_size, kind, a, b, static_cast<double*>(out) + at, (double)bias);
            else ok = sh ? unshuffle_widen_any<int64_t>(p, ne, elem_size, kind, a, b, static_cast<int64_t*>(out) + at, bias)
                         : widen_any<int64_t>(p, elem_size, kind, a, b, static_cast<int64_t*>(out) + at, bias);
            if (!ok) { bad.store(1); break; }
        }
    };
    // (threads and scratch per call: a pool that outlives the call, each thread keeping its buffers, measured level on the
    // 10^9-pixel file - 4.16-4.28 s against 4.01-4.26 s end to end, same box - and is not worth its fork and exit hazards)
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (std::thread& t : pool) t.join();
    return bad.load() ? HPK_ERR_INVALID : HPK_OK;
}
}  // namespace

extern "C" int hpk_decode_chunks(const void* const* src, const uint64_t* src_len, int64_t nchunks, int64_t first_chunk, int64_t chunk_elems,
                                 int32_t elem_size, int32_t kind, int32_t shuffle, int64_t start, int64_t stop, void* out,
                                 int32_t out_f64, int64_t bias, int32_t threads) {
    if (!src) return HPK_ERR_INVALID;
    return decode_impl(src, -1, nullptr, src_len, nchunks, first_chunk, chunk_elems, elem_size, kind, shuffle, start, stop, out, out_f64, bias, threads);
}

extern "C" int hpk_decode_chunks_fd(int32_t fd, const uint64_t* file_off, const uint64_t* src_len, int64_t nchunks, int64_t first_chunk,
                                    int64_t chunk_elems, int32_t elem_size, int32_t kind, int32_t shuffle, int64_t start, int64_t stop,
                                    void* out, int32_t out_f64, int64_t bias, int32_t threads) {
    return decode_impl(nullptr, fd, file_off, src_len, nchunks, first_chunk, chunk_elems, elem_size, kind, shuffle, start, stop, out, out_f64, bias, threads);
}

// The pixels of a chromosome's rows that lie inside the chromosome (bin2 below its bin count: cooler lists a row's trans pixels
// too) and, for files that store both triangles, on or above the diagonal - what `keep = b2 < n; b1[keep], ...` did in numpy on one
// thread, three boolean-mask copies of 0.8 GB columns for a large chromosome of a real map.  Two passes on `threads` threads:
// kept pixels per block, then every block to its place in the output (which must not overlap the input).  Returns the number kept
// (== n, or no outputs given: nothing was written) or a negative status.
extern "C" int64_t hpk_compact_pixels(const int64_t* bin1, const int64_t* bin2, const void* count, int32_t count_size, int64_t n,
                                      int64_t nbins, int32_t square, int64_t* out1, int64_t* out2, void* outc, int32_t threads) {
    if (n < 0 || (n > 0 && (!bin1 || !bin2 || !count)) || (count_size != 4 && count_size != 8)) return HPK_ERR_INVALID;
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(threads > 0 ? threads : (int)std::thread::hardware_concurrency(), (n + (1 << 16) - 1) >> 16));
    const int64_t blk = (n + nt - 1) / std::max(nt, 1);
    std::vector<int64_t> kept((size_t)nt + 1, 0);
    auto keep = [&](int64_t i) { return bin2[i] < nbins && bin2[i] >= 0 && (!square || bin2[i] >= bin1[i]); };
    auto each = [&](const std::function<void(int)>& f) {
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; ++t) pool.emplace_back(f, t);
        f(0);
        for (std::thread& t : pool) t.join();
    };
    each([&](int t) {
        int64_t k = 0;
        for (int64_t i = t * blk, e = std::min(n, (t + 1) * blk); i < e; ++i) k += keep(i) ? 1 : 0;
        kept[(size_t)t + 1] = k;
    });
    for (int t = 0; t < nt; ++t) kept[(size_t)t + 1] += kept[(size_t)t];
    const int64_t total = kept[(size_t)nt];
    if (total == n || (!out1 && !out2 && !outc)) return total;      // (no outputs: the count alone)
    if (!out1 || !out2 || !outc) return HPK_ERR_INVALID;
    each([&](int t) {
        int64_t o = kept[(size_t)t];
        for (int64_t i = t * blk, e = std::min(n, (t + 1) * blk); i < e; ++i) {
            if (!keep(i)) continue;
            out1[o] = bin1[i]; out2[o] = bin2[i];
            if (count_size == 4) static_cast<int32_t*>(outc)[o] = static_cast<const int32_t*>(count)[i];
            else static_cast<int64_t*>(outc)[o] = static_cast<const int64_t*>(count)[i];      // (f64 counts: moved as 8 bytes)
            ++o;
        }
    });
    return total;
}
