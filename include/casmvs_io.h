/* casmvs_io.h - host-side file decoding for the input pipeline of the cascade-MVS hot path.
 *
 * The reference decodes every view of every sample with PIL (`Image.open(...)`: datasets/dtu.py:168-170 for the
 * images, dtu.py:113-118 for the visibility masks) inside torch DataLoader workers (train.py:85-97).  At the MI355X
 * engine's rate (~900 depth maps/s = ~2 700 images/s of 640 x 512) that decode is the limit of the files -> depth-maps
 * path (a GPU pod is granted 16 host cores).  libcasmvs_io.so is a plain C++ library (no HIP, no
 * torch, no zlib / libpng dependency): its own inflate (RFC 1950 / 1951) and PNG un-filtering (ISO 15948), built by
 * casmvsnet_pl_amd/build.py with g++.  The functions are thread-safe and keep no global state; Python calls them through
 * ctypes (which releases the GIL), one call per image from the loader's threads.
 *
 * Results are byte-identical to `np.asarray(Image.open(f).convert("RGB"))` / `.convert("L")` for the formats the functions
 * accept (non-interlaced, 8 bits per sample: grey, grey + alpha, RGB, RGBA, 8-bit palette); everything else returns
 * CASMVS_IO_UNSUPPORTED and the caller keeps using PIL for that file.
 */
#ifndef CASMVS_IO_H
#define CASMVS_IO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CASMVS_IO_OK 0
#define CASMVS_IO_UNSUPPORTED 1 /* a valid PNG this decoder does not handle (interlaced, 1/2/4/16-bit samples): use PIL */
#define CASMVS_IO_CORRUPT 2     /* not a PNG / damaged stream: casmvs_io_last_error() says where */
#define CASMVS_IO_BAD_ARGUMENT 3

/* Message of the last failing call on the calling thread ("" if none). */
const char *casmvs_io_last_error(void);

/* Header of a PNG file image in memory: width, height, channels of the stored pixels (1 grey, 2 grey + alpha, 3 RGB or
 * palette, 4 RGBA).  Replaces the size query of PIL's lazy `Image.open` (datasets/dtu.py:168). */
int casmvs_png_info(const uint8_t *file, size_t file_bytes, int32_t *width, int32_t *height, int32_t *channels);

/* Decode a PNG file image into `out` (height x width x out_channels uint8, rows `out_row_bytes` apart, >= width *
 * out_channels).  out_channels = 3: what `.convert("RGB")` gives (grey replicated, alpha dropped, palette looked up);
 * out_channels = 1: what `.convert("L")` gives (PIL's ITU-R 601 integer luma for colour sources).
 * Replaces `Image.open(f).convert(...)` of datasets/dtu.py:114,168 (and blendedmvs.py / tanks.py where the files are PNG). */
int casmvs_png_decode(const uint8_t *file, size_t file_bytes, uint8_t *out, size_t out_row_bytes, int32_t out_channels);

/* The same from a path (the file is read with one read(2)); width / height must match the caller's buffer. */
int casmvs_png_decode_file(const char *path, uint8_t *out, size_t out_row_bytes, int32_t width, int32_t height, int32_t out_channels);

/* Decode `n` PNG files of equal size into out[i] = out + i * image_stride on `threads` native threads (0: one per
 * file, at most the hardware concurrency).  status[i] receives each file's return code; returns the first non-zero one.
 * This is the whole image side of one batch (B samples x V views) in one GIL-free call. */
int casmvs_png_decode_files(const char *const *paths, int32_t n, uint8_t *out, size_t image_stride, size_t out_row_bytes, int32_t width,
                            int32_t height, int32_t out_channels, int32_t threads, int32_t *status);

/* zlib-stream (RFC 1950) decompression into a buffer of known capacity; *out_bytes receives the decoded size.  Exposed for
 * the tests (every stream python's zlib produces must decode to the same bytes). */
int casmvs_zlib_inflate(const uint8_t *src, size_t src_bytes, uint8_t *dst, size_t dst_capacity, size_t *out_bytes);

#ifdef __cplusplus
}
#endif
#endif
