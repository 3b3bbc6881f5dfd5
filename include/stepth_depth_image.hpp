// stepth_depth_image.hpp -- C++ host-side mirror of the reference's `DepthImage` for the
// open_depth_from_additional path, on top of the C-ABI (stepth_amd.h / libstepth_amd.so).
//
// Mirrors /root/reference/src/depth_image.rs:6-160: same method names, argument meaning and
// error behaviour.  Rust's `Result<(), io::Error>` (ErrorKind::InvalidInput with a fixed
// message) becomes `stepth::Result`: ok() or the message "Failed to open image" /
// "Sizes don't match".  All compute of the path runs in the HIP engine; this header only
// holds the buffers and reads/writes binary PNM files (P6 = RGB8, P5 = Luma8) -- no JPEG/PNG
// decoder is linked here, other formats fail with "Failed to open image" exactly like an
// unreadable file does in the reference (src/depth_image.rs:13-17).
//
// The consumers the README's second half uses (Readme.md:18-24) are mirrored too and run in
// the engine: invert_depth, depth_split, slice, select_foreground (src/depth_image.rs:162-247)
// and MaskImage{image, mask}::apply_mask / save (src/mask_image.rs:6-10, :196-198, :205-213).
// highlight_depth, resize and the rest of MaskImage are outside the scope.
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <utility>
#include <vector>

#include "stepth_amd.h"

namespace stepth {

struct Result { // Ok(()) or Err(io::Error::new(InvalidInput, message))
    std::string message;
    bool ok() const { return message.empty(); }
    static Result Ok() { return {}; }
    static Result Err(const char *m) { return {m}; }
};

struct Dimensions { // disage::Dimensions
    uint32_t width = 0, height = 0;
};

// image::ImageBuffer<Rgba<u8>|Luma<u8>, Vec<u8>>: row-major, no padding
struct Buffer {
    uint32_t width = 0, height = 0, channels = 0;
    std::vector<uint8_t> data;
};

namespace detail {
inline bool read_token(FILE *f, long &v)
{
    int c = fgetc(f);
    for (;;) { // whitespace and # comments
        while (c == ' ' || c == '\t' || c == '\n' || c == '\r')
            c = fgetc(f);
        if (c != '#')
            break;
        while (c != '\n' && c != EOF)
            c = fgetc(f);
    }
    if (c < '0' || c > '9')
        return false;
    v = 0;
    while (c >= '0' && c <= '9') {
        v = v * 10 + (c - '0');
        if (v > 1000000)
            return false;
        c = fgetc(f);
    }
    return true; // one whitespace byte after the token has been consumed
}

// binary PNM: P6 -> 3 channels, P5 -> 1 channel, maxval 255
inline bool read_pnm(const std::string &path, Buffer &out)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f)
        return false;
    bool good = false;
    char magic[2] = {0, 0};
    long w = 0, h = 0, maxv = 0;
    if (fread(magic, 1, 2, f) == 2 && magic[0] == 'P' && (magic[1] == '5' || magic[1] == '6') &&
        read_token(f, w) && read_token(f, h) && read_token(f, maxv) && w > 0 && h > 0 &&
        w <= 65535 && h <= 65535 && maxv == 255) {
        out.width = (uint32_t)w;
        out.height = (uint32_t)h;
        out.channels = magic[1] == '6' ? 3u : 1u;
        out.data.resize((size_t)w * h * out.channels);
        good = fread(out.data.data(), 1, out.data.size(), f) == out.data.size();
    }
    fclose(f);
    return good;
}

inline bool write_pnm(const std::string &path, const Buffer &b)
{
    if (b.channels != 1 && b.channels != 3)
        return false;
    FILE *f = fopen(path.c_str(), "wb");
    if (!f)
        return false;
    fprintf(f, "P%c\n%u %u\n255\n", b.channels == 3 ? '6' : '5', b.width, b.height);
    const bool good = fwrite(b.data.data(), 1, b.data.size(), f) == b.data.size();
    return fclose(f) == 0 && good;
}
} // namespace detail

// DynamicImage::to_rgba8 / to_rgb8 / to_luma8 for the two formats read here
inline Buffer to_rgba8(const Buffer &s)
{
    Buffer d{s.width, s.height, 4, std::vector<uint8_t>((size_t)s.width * s.height * 4)};
    for (size_t i = 0, n = (size_t)s.width * s.height; i < n; ++i) {
        const uint8_t *p = &s.data[i * s.channels];
        d.data[4 * i + 0] = p[0];
        d.data[4 * i + 1] = s.channels >= 3 ? p[1] : p[0];
        d.data[4 * i + 2] = s.channels >= 3 ? p[2] : p[0];
        d.data[4 * i + 3] = s.channels == 4 ? p[3] : 255;
    }
    return d;
}

inline Buffer to_rgb8(const Buffer &s)
{
    Buffer d{s.width, s.height, 3, std::vector<uint8_t>((size_t)s.width * s.height * 3)};
    for (size_t i = 0, n = (size_t)s.width * s.height; i < n; ++i) {
        const uint8_t *p = &s.data[i * s.channels];
        d.data[3 * i + 0] = p[0];
        d.data[3 * i + 1] = s.channels >= 3 ? p[1] : p[0];
        d.data[3 * i + 2] = s.channels >= 3 ? p[2] : p[0];
    }
    return d;
}

// what the reference hard-codes or delegates to disage (src/depth_image.rs:101-110, :119)
struct DepthOptions {
    uint32_t max_steps = 255;         // the literal at :119
    uint32_t tile_w = 1, tile_h = 1;  // uniform groups when `groups` is empty
    std::vector<stepth_group> groups; // stand-in for disage::open::rgb_discrete's output
    bool adaptive = false;            // groups from stepth_adaptive_groups with the reference's
                                      // (16, ceil(log2(W*H))) splits (:101-103); own design
    bool gaussian_tail = true;        // the same-size Gaussian resample at :133-134
};

// `Option<u8>` of depth_split's result and slice's arguments
struct OptionU8 {
    bool some = false;
    uint8_t value = 0;
    static OptionU8 Some(uint8_t v) { return {true, v}; }
    static OptionU8 None() { return {}; }
    bool operator==(const OptionU8 &o) const { return some == o.some && (!some || value == o.value); }
};
using Zone = std::pair<OptionU8, OptionU8>; // (Option<u8>, Option<u8>), depth_image.rs:162

// `pub struct MaskImage { pub image: RgbaImage, pub mask: GrayImage }`, mask_image.rs:6-10 --
// only what DepthImage::slice returns and the README flow uses
class MaskImage {
  public:
    Buffer image; // Rgba8
    Buffer mask;  // Luma8, MASK_TRUE = 255 / MASK_FALSE = 0 (mask_image.rs:3-4)

    uint32_t width() const { return image.width; }
    uint32_t height() const { return image.height; }

    // src/mask_image.rs:205-213
    Result apply_mask(stepth_engine *engine)
    {
        stepth_image ri{image.data.data(), width(), height(), 4, STEPTH_MEM_HOST, (size_t)width() * 4};
        stepth_image mi{mask.data.data(), mask.width, mask.height, 1, STEPTH_MEM_HOST, (size_t)mask.width};
        const int rc = stepth_apply_mask(engine, &ri, &mi, 0);
        return rc == STEPTH_OK ? Result::Ok() : Result::Err(stepth_last_error(engine));
    }

    // src/mask_image.rs:196-198 (binary PPM: the RGB planes of `image`)
    bool save(const std::string &path) const { return detail::write_pnm(path, to_rgb8(image)); }
};

// `pub struct DepthImage { pub image: RgbaImage, pub depth: GrayImage }`, depth_image.rs:6-10
class DepthImage {
  public:
    Buffer image; // Rgba8
    Buffer depth; // Luma8

    using Options = DepthOptions;

    DepthImage() = default; // #[derive(Default)]

    // src/depth_image.rs:12-21
    static Result open(const std::string &image_path, DepthImage &out)
    {
        Buffer img;
        if (!detail::read_pnm(image_path, img))
            return Result::Err("Failed to open image");
        out = from_image(img);
        return Result::Ok();
    }

    // src/depth_image.rs:23-27
    static DepthImage from_image(const Buffer &img)
    {
        DepthImage d;
        d.image = to_rgba8(img);
        d.depth = Buffer{img.width, img.height, 1,
                         std::vector<uint8_t>((size_t)img.width * img.height, 0)};
        return d;
    }

    uint32_t width() const { return image.width; }    // :138-140
    uint32_t height() const { return image.height; }  // :142-144
    Dimensions dimensions() const { return {width(), height()}; } // :155-160

    // src/depth_image.rs:37-49
    Result load_depth(Buffer d)
    {
        if (d.channels == 1 && d.width == width() && d.height == height()) {
            depth = std::move(d);
            return Result::Ok();
        }
        return Result::Err("Sizes don't match");
    }

    // src/depth_image.rs:65-74
    Result open_depth(const std::string &depth_path)
    {
        Buffer d;
        if (!detail::read_pnm(depth_path, d) || d.channels != 1)
            return Result::Err("Failed to open image");
        return load_depth(std::move(d));
    }

    // src/depth_image.rs:76-89
    Result open_depth_from_additional(stepth_engine *engine, const std::string &add_path,
                                      const uint8_t precision[3], const Options &opt = Options())
    {
        Buffer add;
        if (!detail::read_pnm(add_path, add))
            return Result::Err("Failed to open image");
        return load_depth_from_additional(engine, add, precision, opt);
    }

    // src/depth_image.rs:91-136.  Where the reference panics on a zero maximum (:127) this
    // returns Err("every distance is zero ...") and leaves `depth` all zero.
    Result load_depth_from_additional(stepth_engine *engine, const Buffer &add_image,
                                      const uint8_t precision[3], const Options &opt = Options())
    {
        const Buffer add = to_rgb8(add_image); // :96
        Buffer out{width(), height(), 1, std::vector<uint8_t>((size_t)width() * height(), 0)};
        stepth_image mi{image.data.data(), width(), height(), 4, STEPTH_MEM_HOST,
                        (size_t)width() * 4};
        stepth_image ai{add.data.data(), add.width, add.height, 3, STEPTH_MEM_HOST,
                        (size_t)add.width * 3};
        stepth_image di{out.data.data(), out.width, out.height, 1, STEPTH_MEM_HOST,
                        (size_t)out.width};
        stepth_params p{};
        p.struct_size = sizeof p;
        p.mode = STEPTH_MODE_RING_FIRST_HIT;
        p.precision[0] = precision[0];
        p.precision[1] = precision[1];
        p.precision[2] = precision[2];
        p.max_steps = opt.max_steps;
        p.tile_w = opt.tile_w;
        p.tile_h = opt.tile_h;
        // (photographs: most pixels match nearby, a few search far -- STEPTH_FLAG_HEAVY_FIRST)
        p.flags = (opt.gaussian_tail ? STEPTH_FLAG_GAUSSIAN_TAIL : 0u) | STEPTH_FLAG_HEAVY_FIRST;
        uint32_t max = 0;
        std::vector<stepth_group> adaptive;
        if (opt.adaptive && opt.groups.empty()) {
            adaptive.resize((size_t)width() * height());
            size_t n = 0;
            const int grc = stepth_adaptive_groups(engine, &mi, precision, 16, 0, adaptive.data(),
                                                   adaptive.size(), STEPTH_MEM_HOST, &n);
            if (grc != STEPTH_OK)
                return Result::Err(stepth_last_error(engine));
            adaptive.resize(n);
        }
        const std::vector<stepth_group> &gl = adaptive.empty() ? opt.groups : adaptive;
        const int rc = stepth_depth_from_additional(
            engine, &mi, &ai, &p, gl.empty() ? nullptr : gl.data(), gl.size(), STEPTH_MEM_HOST,
            &di, &max);
        if (rc == STEPTH_ERR_MAX_ZERO) {
            depth = std::move(out);
            return Result::Err(stepth_status_string(rc));
        }
        if (rc != STEPTH_OK)
            return Result::Err(rc == STEPTH_ERR_SIZE_MISMATCH ? "Sizes don't match"
                                                              : stepth_last_error(engine));
        return load_depth(std::move(out)); // :135
    }

    // ---- consumers of a finished depth map ------------------------------------------------
    // src/depth_image.rs:226-228
    Result invert_depth(stepth_engine *engine)
    {
        const stepth_image di = depth_view();
        const int rc = stepth_invert_depth(engine, &di, 0);
        return rc == STEPTH_OK ? Result::Ok() : Result::Err(stepth_last_error(engine));
    }

    // src/depth_image.rs:162-219.  Where the reference panics (:203-217) this returns Err with
    // the status text and leaves `zones` empty.
    Result depth_split(stepth_engine *engine, uint8_t n_zones, std::vector<Zone> &zones) const
    {
        zones.clear();
        stepth_zone z[256];
        uint32_t n = 0;
        const stepth_image di = depth_view();
        const int rc = stepth_depth_split(engine, &di, n_zones, z, 256, &n);
        if (rc != STEPTH_OK)
            return Result::Err(rc <= STEPTH_ERR_SPLIT_STEP && rc >= STEPTH_ERR_SPLIT_DIVERGED
                                   ? stepth_status_string(rc)
                                   : stepth_last_error(engine));
        for (uint32_t i = 0; i < n; ++i)
            zones.push_back({z[i].has_from ? OptionU8::Some(z[i].from) : OptionU8::None(),
                             z[i].has_to ? OptionU8::Some(z[i].to) : OptionU8::None()});
        return Result::Ok();
    }

    // src/depth_image.rs:230-247
    Result slice(stepth_engine *engine, OptionU8 from, OptionU8 to, MaskImage &out) const
    {
        out.image = image; // :243 self.image.clone()
        out.mask = Buffer{width(), height(), 1, std::vector<uint8_t>((size_t)width() * height(), 0)};
        const stepth_zone range{(uint8_t)from.some, from.value, (uint8_t)to.some, to.value};
        const stepth_image di = depth_view();
        stepth_image mi{out.mask.data.data(), width(), height(), 1, STEPTH_MEM_HOST, (size_t)width()};
        const int rc = stepth_slice(engine, &di, &range, &mi, 0);
        return rc == STEPTH_OK ? Result::Ok() : Result::Err(stepth_last_error(engine));
    }

    // src/depth_image.rs:221-224
    Result select_foreground(stepth_engine *engine, MaskImage &out) const
    {
        out.image = image;
        out.mask = Buffer{width(), height(), 1, std::vector<uint8_t>((size_t)width() * height(), 0)};
        const stepth_image di = depth_view();
        stepth_image mi{out.mask.data.data(), width(), height(), 1, STEPTH_MEM_HOST, (size_t)width()};
        const int rc = stepth_select_foreground(engine, &di, &mi, nullptr);
        if (rc == STEPTH_OK)
            return Result::Ok();
        return Result::Err(rc <= STEPTH_ERR_SPLIT_STEP && rc >= STEPTH_ERR_SPLIT_DIVERGED
                               ? stepth_status_string(rc)
                               : stepth_last_error(engine));
    }

  private:
    stepth_image depth_view() const
    {
        return stepth_image{depth.data.data(), depth.width, depth.height, 1, STEPTH_MEM_HOST,
                            (size_t)depth.width};
    }
};

} // namespace stepth
