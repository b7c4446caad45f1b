// codecs.hpp -- zstd, xz and bzip2 streams for the command-line driver (get_writer, src/local_filter.rs:110-151; niffler's
// format sniffing on input, :41-55).  This image ships the RUNTIME libraries (libzstd.so.1, liblzma.so.5) but not
// their headers, so the few entry points used are declared here exactly as the libraries' stable C ABIs define them
// and bound with dlopen when a .zst / .xz stream is first met; a machine without the library gets a clear error
// instead of a guess.  gzip stays on zlib (linked directly).
#ifndef DEACON_HIP_CODECS_HPP
#define DEACON_HIP_CODECS_HPP

#include <dlfcn.h>
#include <stddef.h>
#include <stdint.h>

#include <string>

namespace codecs {

// ---- zstd (stable since v1.0: lib/zstd.h, "Streaming" section) -------------------------------------------------------
struct ZSTD_inBuffer {
    const void *src;
    size_t size;
    size_t pos;
};
struct ZSTD_outBuffer {
    void *dst;
    size_t size;
    size_t pos;
};
struct Zstd {
    void *(*createDStream)();
    size_t (*initDStream)(void *);
    size_t (*decompressStream)(void *, ZSTD_outBuffer *, ZSTD_inBuffer *);
    size_t (*freeDStream)(void *);
    void *(*createCStream)();
    size_t (*initCStream)(void *, int level);
    size_t (*compressStream)(void *, ZSTD_outBuffer *, ZSTD_inBuffer *);
    size_t (*endStream)(void *, ZSTD_outBuffer *);
    size_t (*freeCStream)(void *);
    unsigned (*isError)(size_t);
    const char *(*getErrorName)(size_t);
    static const Zstd *get(std::string *err) {
        static Zstd api;
        static int state = 0;  // 0 = not tried, 1 = ok, -1 = unavailable
        static std::string why;
        if (state == 0) {
            void *h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
            if (!h) h = dlopen("libzstd.so", RTLD_NOW | RTLD_LOCAL);
            auto sym = [&](const char *n) {
                void *p = h ? dlsym(h, n) : nullptr;
                if (!p && why.empty()) why = h ? std::string("libzstd has no ") + n : std::string("libzstd.so.1 not found");
                return p;
            };
            api.createDStream = (void *(*)())sym("ZSTD_createDStream");
            api.initDStream = (size_t(*)(void *))sym("ZSTD_initDStream");
            api.decompressStream = (size_t(*)(void *, ZSTD_outBuffer *, ZSTD_inBuffer *))sym("ZSTD_decompressStream");
            api.freeDStream = (size_t(*)(void *))sym("ZSTD_freeDStream");
            api.createCStream = (void *(*)())sym("ZSTD_createCStream");
            api.initCStream = (size_t(*)(void *, int))sym("ZSTD_initCStream");
            api.compressStream = (size_t(*)(void *, ZSTD_outBuffer *, ZSTD_inBuffer *))sym("ZSTD_compressStream");
            api.endStream = (size_t(*)(void *, ZSTD_outBuffer *))sym("ZSTD_endStream");
            api.freeCStream = (size_t(*)(void *))sym("ZSTD_freeCStream");
            api.isError = (unsigned (*)(size_t))sym("ZSTD_isError");
            api.getErrorName = (const char *(*)(size_t))sym("ZSTD_getErrorName");
            state = why.empty() ? 1 : -1;
        }
        if (state < 0) {
            if (err) *err = why;
            return nullptr;
        }
        return &api;
    }
};

// ---- xz / liblzma 5.x (src/liblzma/api/lzma/base.h, container.h) ---------------------------------------------------------
struct lzma_stream {
    const uint8_t *next_in;
    size_t avail_in;
    uint64_t total_in;
    uint8_t *next_out;
    size_t avail_out;
    uint64_t total_out;
    const void *allocator;
    void *internal;
    void *reserved_ptr1, *reserved_ptr2, *reserved_ptr3, *reserved_ptr4;
    uint64_t reserved_int1, reserved_int2;
    size_t reserved_int3, reserved_int4;
    int reserved_enum1, reserved_enum2;
};
enum { LZMA_OK = 0, LZMA_STREAM_END = 1, LZMA_RUN = 0, LZMA_FINISH = 3, LZMA_CHECK_CRC64 = 4, LZMA_CONCATENATED = 0x08 };
struct Lzma {
    int (*stream_decoder)(lzma_stream *, uint64_t memlimit, uint32_t flags);
    int (*easy_encoder)(lzma_stream *, uint32_t preset, int check);
    int (*code)(lzma_stream *, int action);
    void (*end)(lzma_stream *);
    static const Lzma *get(std::string *err) {
        static Lzma api;
        static int state = 0;
        static std::string why;
        if (state == 0) {
            void *h = dlopen("liblzma.so.5", RTLD_NOW | RTLD_LOCAL);
            if (!h) h = dlopen("liblzma.so", RTLD_NOW | RTLD_LOCAL);
            auto sym = [&](const char *n) {
                void *p = h ? dlsym(h, n) : nullptr;
                if (!p && why.empty()) why = h ? std::string("liblzma has no ") + n : std::string("liblzma.so.5 not found");
                return p;
            };
            api.stream_decoder = (int (*)(lzma_stream *, uint64_t, uint32_t))sym("lzma_stream_decoder");
            api.easy_encoder = (int (*)(lzma_stream *, uint32_t, int))sym("lzma_easy_encoder");
            api.code = (int (*)(lzma_stream *, int))sym("lzma_code");
            api.end = (void (*)(lzma_stream *))sym("lzma_end");
            state = why.empty() ? 1 : -1;
        }
        if (state < 0) {
            if (err) *err = why;
            return nullptr;
        }
        return &api;
    }
};

// ---- bzip2 1.0 (bzlib.h): input only -- the reference's reader sniffs it too (niffler's default formats), its writer has no
// such extension (get_writer, src/local_filter.rs:110-151)
struct bz_stream {
    char *next_in;
    unsigned int avail_in, total_in_lo32, total_in_hi32;
    char *next_out;
    unsigned int avail_out, total_out_lo32, total_out_hi32;
    void *state;
    void *(*bzalloc)(void *, int, int);
    void (*bzfree)(void *, void *);
    void *opaque;
};
enum { BZ_OK = 0, BZ_STREAM_END = 4 };
struct Bz2 {
    int (*decompress_init)(bz_stream *, int verbosity, int small);
    int (*decompress)(bz_stream *);
    int (*decompress_end)(bz_stream *);
    static const Bz2 *get(std::string *err) {
        static Bz2 api;
        static int state = 0;
        static std::string why;
        if (state == 0) {
            void *h = dlopen("libbz2.so.1.0", RTLD_NOW | RTLD_LOCAL);
            if (!h) h = dlopen("libbz2.so.1", RTLD_NOW | RTLD_LOCAL);
            if (!h) h = dlopen("libbz2.so", RTLD_NOW | RTLD_LOCAL);
            auto sym = [&](const char *n) {
                void *p = h ? dlsym(h, n) : nullptr;
                if (!p && why.empty()) why = h ? std::string("libbz2 has no ") + n : std::string("libbz2.so.1.0 not found");
                return p;
            };
            api.decompress_init = (int (*)(bz_stream *, int, int))sym("BZ2_bzDecompressInit");
            api.decompress = (int (*)(bz_stream *))sym("BZ2_bzDecompress");
            api.decompress_end = (int (*)(bz_stream *))sym("BZ2_bzDecompressEnd");
            state = why.empty() ? 1 : -1;
        }
        if (state < 0) {
            if (err) *err = why;
            return nullptr;
        }
        return &api;
    }
};

inline bool is_bz2_magic(const unsigned char *p, size_t n) { return n >= 4 && p[0] == 'B' && p[1] == 'Z' && p[2] == 'h' && p[3] >= '1' && p[3] <= '9'; }
inline bool is_zstd_magic(const unsigned char *p, size_t n) { return n >= 4 && p[0] == 0x28 && p[1] == 0xB5 && p[2] == 0x2F && p[3] == 0xFD; }
inline bool is_xz_magic(const unsigned char *p, size_t n) {
    return n >= 6 && p[0] == 0xFD && p[1] == '7' && p[2] == 'z' && p[3] == 'X' && p[4] == 'Z' && p[5] == 0x00;
}
inline bool is_gzip_magic(const unsigned char *p, size_t n) { return n >= 2 && p[0] == 0x1F && p[1] == 0x8B; }

}  // namespace codecs
#endif
