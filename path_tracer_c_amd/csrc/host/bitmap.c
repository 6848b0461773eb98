/*
 * bitmap.c -- 24-bit uncompressed BMP writer.  File layout follows the
 * reference's src/bitmap.c:5-78: 14-byte file header + 40-byte
 * BITMAPINFOHEADER, positive height (so the first stored row is the bottom of
 * the picture), BGR byte order, rows padded to 4 bytes.  `y_inverted` says
 * the caller's row 0 is the TOP of the picture (the CPU/GPU frame layout,
 * src/renderer.c:111-114), so rows are emitted last-to-first.
 *
 * The header is assembled byte by byte (little-endian by construction, no
 * <endian.h>), and each row is written with one fwrite.
 */
#include "path_tracer_api.h"

#include <stdio.h>
#include <string.h>

static void put_u16(uint8_t *at, uint32_t v) {
    at[0] = (uint8_t)(v & 0xFF);
    at[1] = (uint8_t)((v >> 8) & 0xFF);
}

static void put_u32(uint8_t *at, uint32_t v) {
    put_u16(at, v & 0xFFFF);
    put_u16(at + 2, v >> 16);
}

void write_bitmap(char *filename, unsigned int imgwidth, unsigned int imgheight,
                  uint8_t *pixels, bool y_inverted) {
    const uint32_t row_bytes = 3u * imgwidth;
    const uint32_t padding = (4u - row_bytes % 4u) % 4u;
    const uint32_t stride = row_bytes + padding;
    const uint32_t image_bytes = stride * imgheight;

    FILE *out = fopen(filename, "wb");
    if (out == NULL) {
        perror(filename);
        return;
    }

    uint8_t header[54];
    memset(header, 0, sizeof header);
    header[0] = 'B';
    header[1] = 'M';
    put_u32(header + 2, 54u + image_bytes); /* file size */
    put_u32(header + 10, 54u);              /* offset of pixel data */
    put_u32(header + 14, 40u);              /* info header size */
    put_u32(header + 18, imgwidth);
    put_u32(header + 22, imgheight);
    put_u16(header + 26, 1u);  /* planes */
    put_u16(header + 28, 24u); /* bits per pixel */
    put_u32(header + 34, image_bytes);
    fwrite(header, 1, sizeof header, out);

    uint8_t *line = calloc(stride ? stride : 1, 1);
    if (line == NULL) {
        fclose(out);
        return;
    }
    for (uint32_t k = 0; k < imgheight; k++) {
        uint32_t y = y_inverted ? imgheight - 1u - k : k;
        const uint8_t *src = pixels + (size_t)y * row_bytes;
        for (uint32_t x = 0; x < imgwidth; x++) {
            line[3 * x + 0] = src[3 * x + 2]; /* B */
            line[3 * x + 1] = src[3 * x + 1]; /* G */
            line[3 * x + 2] = src[3 * x + 0]; /* R */
        }
        fwrite(line, 1, stride, out);
    }
    free(line);
    fclose(out);
}
