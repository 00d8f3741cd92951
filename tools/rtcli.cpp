// rtcli — command-line renderer: the counterpart of the reference's src/bin/main.rs on top of the GPU backend.
//
//   rtcli <light-sampler 1..4> [--scene head] [--width 800] [--height 800] [--spp 800] [--shader mis|path|whitted|normal]
//         [--pixel-sampler 1..4] [--seed 1] [--device 0] [--assets assets] [--out ./output/out.ppm] [--batch N]
//         [--resume FILE] [--stop-after N] [--upscale RATE]
//
// argv[1] keeps its meaning (SamplerKind::from_int for the light sampler, main.rs:15-18); the compile-time constants
// NX, NY, NS of src/lib.rs:21-23 and the scene / shader chosen by (un)commenting code (main.rs:20-23,40-43) are
// flags.  Output: the same ASCII P3 file (main.rs:27,51) and the same progress / "Finished." lines (main.rs:53-61),
// printed per batch of samples instead of per five rows because the image is rendered a sample range at a time.
// An --out name ending in .png gives an 8-bit PNG instead; --upscale enlarges the image by pixel repetition first
// (the job of resize.py for the sampler comparison images).

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/rt_api.h"
#include "../include/rt_host.h"

static int usage() {
    std::fprintf(stderr, "usage: rtcli <light-sampler 1..4> [--scene NAME] [--width W] [--height H] [--spp N] [--shader mis|path|whitted|normal]\n"
                         "             [--pixel-sampler 1..4] [--seed S] [--device D] [--assets DIR] [--out FILE] [--batch N]\n"
                         "             [--resume FILE: sums + samples done, saved after every batch, picked up if present]\n"
                         "             [--stop-after N: leave after N samples per pixel (with --resume: continue later)]\n"
                         "             [--upscale RATE: repeat every pixel RATE x RATE times]   (--out FILE.png writes a PNG)\n");
    return 2;
}

int main(int argc, char **argv) {
    auto now = std::chrono::steady_clock::now();
    if (argc < 2) return usage();
    int light_sampler = std::atoi(argv[1]);
    std::string scene = "head", shader = "mis", assets = "assets", out = "./output/out.ppm", resume;
    int width = 800, height = 800, spp = 800, pixel_sampler = RT_SAMPLER_BLUE, device = 0, batch = 0, stop_after = 0, upscale = 1;
    unsigned seed = 1;
    for (int i = 2; i + 1 < argc; i += 2) {
        std::string k = argv[i], v = argv[i + 1];
        if (k == "--scene") scene = v;
        else if (k == "--width") width = std::atoi(v.c_str());
        else if (k == "--height") height = std::atoi(v.c_str());
        else if (k == "--spp") spp = std::atoi(v.c_str());
        else if (k == "--shader") shader = v;
        else if (k == "--pixel-sampler") pixel_sampler = std::atoi(v.c_str());
        else if (k == "--seed") seed = (unsigned)std::strtoul(v.c_str(), nullptr, 10);
        else if (k == "--device") device = std::atoi(v.c_str());
        else if (k == "--assets") assets = v;
        else if (k == "--out") out = v;
        else if (k == "--batch") batch = std::atoi(v.c_str());
        else if (k == "--resume") resume = v;
        else if (k == "--stop-after") stop_after = std::atoi(v.c_str());
        else if (k == "--upscale") upscale = std::atoi(v.c_str());
        else return usage();
    }
    if (light_sampler < 1 || light_sampler > 4) { // SamplerKind::from_int, sampler.rs:13-21
        std::fprintf(stderr, "Error: unexpected integer\n");
        return 1;
    }

    rth_scene *hs = nullptr;
    if (rth_scene_create(scene.c_str(), (float)width / (float)height, light_sampler, assets.c_str(), &hs) != RT_OK) {
        std::fprintf(stderr, "Error: %s\n", rth_last_error());
        return 1;
    }
    rt_context *ctx = nullptr;
    rt_scene *sc = nullptr;
    if (rt_context_create(device, &ctx) != RT_OK || rt_scene_create(ctx, rth_scene_desc(hs), &sc) != RT_OK) {
        std::fprintf(stderr, "Error: %s\n", rt_last_error());
        return 1;
    }

    rt_params p;
    rt_params_default(&p, width, height, spp);
    p.pixel_sampler = pixel_sampler;
    p.seed = seed;
    p.shader = shader == "normal" ? RT_SHADER_NORMAL : shader == "whitted" ? RT_SHADER_WHITTED : shader == "path" ? RT_SHADER_PATH : RT_SHADER_MIS;

    std::vector<float> accum((size_t)width * height * 3, 0.0f);
    if (batch <= 0) batch = spp < 16 ? spp : (spp + 15) / 16;
    unsigned long long rays = 0;
    int done = 0;
    if (!resume.empty()) {
        if (FILE *probe = std::fopen(resume.c_str(), "rb")) {
            std::fclose(probe);
            if (rth_accum_load(resume.c_str(), width, height, spp, seed, &done, accum.data()) != RT_OK) {
                std::fprintf(stderr, "Error: %s\n", rth_last_error());
                return 1;
            }
            std::printf("Resuming at sample %d of %d\n", done, spp);
        }
    }
    const int last = stop_after > 0 && stop_after < spp ? stop_after : spp;
    for (int b = done; b < last; b += batch) {
        p.spp_begin = b;
        p.spp_end = b + batch < last ? b + batch : last;
        rt_stats st;
        if (rt_render(sc, rth_scene_camera(hs), &p, accum.data(), 0, &st) != RT_OK) {
            std::fprintf(stderr, "Error: %s\n", rt_last_error());
            return 1;
        }
        rays += st.rays_closest + st.rays_shadow;
        double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - now).count();
        std::printf("Now rendering: %g%%, %lld seconds elapsed\n", (double)(p.spp_end * 100) / (double)spp, (long long)secs);
        if (!resume.empty() && rth_accum_save(resume.c_str(), width, height, spp, p.spp_end, seed, accum.data()) != RT_OK) {
            std::fprintf(stderr, "Error: %s\n", rth_last_error());
            return 1;
        }
    }
    if (last < spp) {
        std::printf("Stopped after %d of %d samples.\n", last, spp);
        rt_scene_destroy(sc);
        rt_context_destroy(ctx);
        rth_scene_destroy(hs);
        return 0;
    }
    std::vector<int32_t> rgb((size_t)width * height * 3);
    rt_resolve(accum.data(), (int64_t)width * height, spp, rgb.data(), nullptr);
    if (upscale > 1) {
        std::vector<int32_t> big((size_t)width * upscale * height * upscale * 3);
        if (rth_upscale_nearest(width, height, upscale, rgb.data(), big.data()) != RT_OK) {
            std::fprintf(stderr, "Error: %s\n", rth_last_error());
            return 1;
        }
        rgb.swap(big);
        width *= upscale;
        height *= upscale;
    }
    const bool png = out.size() > 4 && out.compare(out.size() - 4, 4, ".png") == 0;
    if ((png ? rth_write_png(out.c_str(), width, height, rgb.data()) : rth_write_ppm(out.c_str(), width, height, rgb.data())) != RT_OK) {
        std::fprintf(stderr, "Error: %s\n", rth_last_error());
        return 1;
    }
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - now).count();
    std::printf("Finished. %llu rays in %.3f s (%.1f Mrays/s)\n", rays, secs, rays / secs / 1e6);
    rt_scene_destroy(sc);
    rt_context_destroy(ctx);
    rth_scene_destroy(hs);
    return 0;
}
