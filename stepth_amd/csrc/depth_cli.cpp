// depth_cli.cpp -- the README flow of the reference (Readme.md:11-16) in C++ on top of the
// host mirror include/stepth_depth_image.hpp:
//     let mut img = DepthImage::open("mains.jpg")?;
//     img.open_depth_from_additional("additionals.jpg", [u8::MAX/7;3])?;
//     img.depth().save("depth.jpg")?;
// and its second half (Readme.md:18-24):
//     img.open_depth("depth.jpg")?; img.invert_depth();
//     let mut mask = img.select_foreground(); mask.apply_mask(); mask.save("foreground.jpg")?;
// with binary PNM files in place of JPEG.
//   stepth_depth_cli main.ppm additional.ppm out.pgm [precision] [--no-gaussian]
//                    [--max-steps N] [--tile W H] [--adaptive]
//   stepth_depth_cli --foreground main.ppm depth.pgm out.ppm [--zones N]
//                    (--zones N prints depth_split(N) of the inverted depth map as well)
//   stepth_depth_cli --check-errors      (error-message contract of the mirror, needs a GPU)
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/stepth_depth_image.hpp"

using stepth::DepthImage;

static int fail(const char *what, const std::string &msg)
{
    fprintf(stderr, "%s: %s\n", what, msg.c_str());
    return 1;
}

static int check_errors(stepth_engine *e)
{
    DepthImage img;
    if (DepthImage::open("/nonexistent/main.ppm", img).message != "Failed to open image")
        return fail("open", "wrong message");
    stepth::Buffer rgb{5, 4, 3, std::vector<uint8_t>(60, 100)};
    img = DepthImage::from_image(rgb);
    if (img.width() != 5 || img.height() != 4 || img.image.channels != 4 || img.image.data[3] != 255 ||
        img.depth.data.size() != 20 || img.dimensions().width != 5)
        return fail("from_image", "wrong buffers");
    if (img.load_depth(stepth::Buffer{4, 5, 1, std::vector<uint8_t>(20)}).message != "Sizes don't match")
        return fail("load_depth", "wrong message");
    if (!img.load_depth(stepth::Buffer{5, 4, 1, std::vector<uint8_t>(20, 7)}).ok() || img.depth.data[0] != 7)
        return fail("load_depth", "rejected a matching buffer");
    const uint8_t p[3] = {36, 36, 36};
    if (img.open_depth_from_additional(e, "/nonexistent/add.ppm", p).message != "Failed to open image")
        return fail("open_depth_from_additional", "wrong message");
    if (img.open_depth("/nonexistent/depth.pgm").message != "Failed to open image")
        return fail("open_depth", "wrong message");
    // flat pair: every distance 0 -> the reference would divide by zero (:127)
    stepth::Result r = img.load_depth_from_additional(e, rgb, p);
    if (r.ok() || img.depth.data[0] != 0)
        return fail("max == 0", "expected an error and an all-zero depth");
    // depth_split on a flat map: the reference panics in step_by (:217) -> Err, no zones
    std::vector<stepth::Zone> zones;
    if (img.depth_split(e, 2, zones).ok() || !zones.empty())
        return fail("depth_split", "expected an error on a flat depth map");
    if (!img.depth_split(e, 1, zones).ok() || zones.size() != 1 || zones[0].first.some || zones[0].second.some)
        return fail("depth_split", "zones < 2 must give (None, None)");
    stepth::MaskImage mask;
    if (img.select_foreground(e, mask).ok())
        return fail("select_foreground", "expected an error on a flat depth map");
    if (!img.slice(e, stepth::OptionU8::None(), stepth::OptionU8::None(), mask).ok() ||
        mask.mask.data.size() != 20 || mask.mask.data[0] != 255 || mask.image.data != img.image.data)
        return fail("slice", "(None, None) must give an all-true mask and a copy of the image");
    printf("OK\n");
    return 0;
}

// Readme.md:18-24
static int foreground(stepth_engine *e, int argc, char **argv)
{
    if (argc < 5) {
        fprintf(stderr, "usage: %s --foreground main.ppm depth.pgm out.ppm [--zones N]\n", argv[0]);
        return 2;
    }
    DepthImage img;
    stepth::Result r = DepthImage::open(argv[2], img);
    if (r.ok())
        r = img.open_depth(argv[3]);
    if (r.ok())
        r = img.invert_depth(e);
    if (!r.ok())
        return fail("stepth", r.message);
    if (argc >= 7 && !strcmp(argv[5], "--zones")) {
        std::vector<stepth::Zone> zones;
        r = img.depth_split(e, (uint8_t)atoi(argv[6]), zones);
        if (!r.ok())
            return fail("depth_split", r.message);
        for (const stepth::Zone &z : zones) {
            if (z.first.some)
                printf("zone %u %u\n", z.first.value, z.second.value);
            else
                printf("zone None None\n");
        }
    }
    stepth::MaskImage mask;
    r = img.select_foreground(e, mask);
    if (r.ok())
        r = mask.apply_mask(e);
    if (!r.ok())
        return fail("select_foreground", r.message);
    if (!mask.save(argv[4]))
        return fail("save", argv[4]);
    return 0;
}

int main(int argc, char **argv)
{
    stepth_engine *e = nullptr;
    const int rc = stepth_engine_create(-1, &e);
    if (rc != STEPTH_OK)
        return fail("stepth_engine_create", stepth_status_string(rc));
    int status = 0;
    if (argc >= 2 && !strcmp(argv[1], "--check-errors")) {
        status = check_errors(e);
    } else if (argc >= 2 && !strcmp(argv[1], "--foreground")) {
        status = foreground(e, argc, argv);
    } else if (argc < 4) {
        fprintf(stderr, "usage: %s main.ppm additional.ppm out.pgm [precision] [--no-gaussian] "
                        "[--max-steps N] [--tile W H] [--adaptive]\n", argv[0]);
        status = 2;
    } else {
        uint8_t prec[3] = {255 / 7, 255 / 7, 255 / 7}; // Readme.md:14
        DepthImage::Options opt;
        for (int i = 4; i < argc; ++i) {
            if (!strcmp(argv[i], "--no-gaussian"))
                opt.gaussian_tail = false;
            else if (!strcmp(argv[i], "--adaptive"))
                opt.adaptive = true;
            else if (!strcmp(argv[i], "--max-steps") && i + 1 < argc)
                opt.max_steps = (uint32_t)atoi(argv[++i]);
            else if (!strcmp(argv[i], "--tile") && i + 2 < argc) {
                opt.tile_w = (uint32_t)atoi(argv[++i]);
                opt.tile_h = (uint32_t)atoi(argv[++i]);
            } else
                prec[0] = prec[1] = prec[2] = (uint8_t)atoi(argv[i]);
        }
        DepthImage img;
        stepth::Result r = DepthImage::open(argv[1], img);
        if (r.ok())
            r = img.open_depth_from_additional(e, argv[2], prec, opt);
        if (!r.ok())
            status = fail("stepth", r.message);
        else if (!stepth::detail::write_pnm(argv[3], img.depth))
            status = fail("save", argv[3]);
    }
    stepth_engine_destroy(e);
    return status;
}
