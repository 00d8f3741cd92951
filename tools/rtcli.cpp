// rtcli — command-line renderer: the counterpart of the reference's src/bin/main.rs on top of the GPU backend.
//
//   rtcli <light-sampler 1..4> [--scene head] [--width 800] [--height 800] [--spp 800] [--shader mis|path|whitted|normal]
//         [--pixel-sampler 1..4] [--seed 1] [--device 0] [--assets assets] [--out ./output/out.ppm] [--batch N]
//         [--resume FILE] [--stop-after N] [--upscale RATE] [--gpus N] [--shard spp|tile]
//
// argv[1] keeps its meaning (SamplerKind::from_int for the light sampler, main.rs:15-18); the compile-time constants
// NX, NY, NS of src/lib.rs:21-23 and the scene / shader chosen by (un)commenting code (main.rs:20-23,40-43) are
// flags.  Output: the same ASCII P3 file (main.rs:27,51) and the same progress / "Finished." lines (main.rs:53-61),
// printed per batch of samples instead of per five rows because the image is rendered a sample range at a time.
// An --out name ending in .png gives an 8-bit PNG instead; --upscale enlarges the image by pixel repetition first
// (the job of resize.py for the sampler comparison images).  --gpus N renders the frame on devices 0..N-1 of this
// process, one host thread and one context per device, split by sample range (--shard spp, the default) or by 64x64
// tiles dealt round-robin (--shard tile); the per-GPU sums meet in rt_reduce_accum (RCCL) on device 0.

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../include/rt_api.h"
#include "../include/rt_host.h"

static int usage() {
    std::fprintf(stderr, "usage: rtcli <light-sampler 1..4> [--scene NAME] [--width W] [--height H] [--spp N] [--shader mis|path|whitted|normal]\n"
                         "             [--pixel-sampler 1..4] [--seed S] [--device D] [--assets DIR] [--out FILE] [--batch N]\n"
                         "             [--resume FILE: sums + samples done, saved after every batch, picked up if present]\n"
                         "             [--stop-after N: leave after N samples per pixel (with --resume: continue later)]\n"
                         "             [--upscale RATE: repeat every pixel RATE x RATE times]   (--out FILE.png writes a PNG)\n"
                         "             [--gpus N: devices 0..N-1, one thread each] [--shard spp|tile: how the frame is split over them]\n");
    return 2;
}

int main(int argc, char **argv) {
    auto now = std::chrono::steady_clock::now();
    if (argc < 2) return usage();
    int light_sampler = std::atoi(argv[1]);
    std::string scene = "head", shader = "mis", assets = "assets", out = "./output/out.ppm", resume, shard = "spp";
    int width = 800, height = 800, spp = 800, pixel_sampler = RT_SAMPLER_BLUE, device = 0, batch = 0, stop_after = 0, upscale = 1, gpus = 1;
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
        else if (k == "--gpus") gpus = std::atoi(v.c_str());
        else if (k == "--shard") shard = v;
        else return usage();
    }
    if (light_sampler < 1 || light_sampler > 4) { // SamplerKind::from_int, sampler.rs:13-21
        std::fprintf(stderr, "Error: unexpected integer\n");
        return 1;
    }

    if (gpus < 1 || (shard != "spp" && shard != "tile")) return usage();
    if (upscale < 1 || (long long)width * upscale > 65535 || (long long)height * upscale > 65535) { // before anything is allocated from it
        std::fprintf(stderr, "Error: --upscale must be >= 1 and keep the image within 65535 pixels per side\n");
        return 2;
    }
    if (gpus > 1 && (!resume.empty() || stop_after > 0 || batch > 0)) {
        std::fprintf(stderr, "Error: --resume, --stop-after and --batch are for one GPU\n");
        return 2;
    }

    rth_scene *hs = nullptr;
    if (rth_scene_create(scene.c_str(), (float)width / (float)height, light_sampler, assets.c_str(), &hs) != RT_OK) {
        std::fprintf(stderr, "Error: %s\n", rth_last_error());
        return 1;
    }
    rt_params p;
    rt_params_default(&p, width, height, spp);
    p.pixel_sampler = pixel_sampler;
    p.seed = seed;
    p.shader = shader == "normal" ? RT_SHADER_NORMAL : shader == "whitted" ? RT_SHADER_WHITTED : shader == "path" ? RT_SHADER_PATH : RT_SHADER_MIS;
    std::vector<float> accum((size_t)width * height * 3, 0.0f);

    if (gpus > 1) {
        // one context per device; a device that does not exist ends the run before anything is rendered
        std::vector<rt_context *> ctxs((size_t)gpus, nullptr);
        for (int g = 0; g < gpus; g++) {
            if (rt_context_create(g, &ctxs[(size_t)g]) != RT_OK) {
                std::fprintf(stderr, "Error: --gpus %d: device %d: %s\n", gpus, g, rt_last_error());
                for (rt_context *c : ctxs) rt_context_destroy(c);
                return 2;
            }
        }
        std::vector<rt_scene *> scs((size_t)gpus, nullptr);
        std::vector<float *> accs((size_t)gpus, nullptr);
        std::vector<rt_stats> sts((size_t)gpus);
        std::vector<std::string> errs((size_t)gpus);
        std::vector<std::thread> threads;
        for (int g = 0; g < gpus; g++) {
            threads.emplace_back([&, g]() { // rt_render blocks: every device is driven by a thread of its own
                rt_params pg = p;
                if (shard == "spp") {
                    pg.spp_begin = (int)((long long)g * spp / gpus);
                    pg.spp_end = (int)((long long)(g + 1) * spp / gpus);
                } else {
                    pg.tile_size = 64;
                    pg.shard_rank = g;
                    pg.shard_count = gpus;
                }
                if (rt_scene_create(ctxs[(size_t)g], rth_scene_desc(hs), &scs[(size_t)g]) != RT_OK ||
                    rt_accum_create(ctxs[(size_t)g], width, height, &accs[(size_t)g]) != RT_OK ||
                    rt_render(scs[(size_t)g], rth_scene_camera(hs), &pg, accs[(size_t)g], 1, &sts[(size_t)g]) != RT_OK)
                    errs[(size_t)g] = rt_last_error(); // the message is the failing thread's own
            });
        }
        for (std::thread &t : threads) t.join();
        for (int g = 0; g < gpus; g++)
            if (!errs[(size_t)g].empty()) {
                std::fprintf(stderr, "Error: device %d: %s\n", g, errs[(size_t)g].c_str());
                return 1;
            }
        if (rt_reduce_accum(ctxs.data(), gpus, accs.data(), (int64_t)width * height * 3, 0) != RT_OK ||
            rt_accum_download(ctxs[0], accs[0], accum.data(), (int64_t)width * height * 3) != RT_OK) {
            std::fprintf(stderr, "Error: %s\n", rt_last_error());
            return 1;
        }
        unsigned long long rays = 0;
        for (const rt_stats &st : sts) rays += st.rays_closest + st.rays_shadow;
        double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - now).count();
        std::printf("Now rendering: 100%%, %lld seconds elapsed\n", (long long)secs);
        std::vector<int32_t> rgb((size_t)width * height * 3);
        rt_resolve(accum.data(), (int64_t)width * height, spp, rgb.data(), nullptr);
        const bool png = out.size() > 4 && out.compare(out.size() - 4, 4, ".png") == 0;
        if ((png ? rth_write_png(out.c_str(), width, height, rgb.data()) : rth_write_ppm(out.c_str(), width, height, rgb.data())) != RT_OK) {
            std::fprintf(stderr, "Error: %s\n", rth_last_error());
            return 1;
        }
        secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - now).count();
        std::printf("Finished. %llu rays on %d GPUs in %.3f s (%.1f Mrays/s)\n", rays, gpus, secs, rays / secs / 1e6);
        for (int g = 0; g < gpus; g++) {
            rt_accum_destroy(ctxs[(size_t)g], accs[(size_t)g]);
            rt_scene_destroy(scs[(size_t)g]);
            rt_context_destroy(ctxs[(size_t)g]);
        }
        rth_scene_destroy(hs);
        return 0;
    }

    rt_context *ctx = nullptr;
    rt_scene *sc = nullptr;
    if (rt_context_create(device, &ctx) != RT_OK || rt_scene_create(ctx, rth_scene_desc(hs), &sc) != RT_OK) {
        std::fprintf(stderr, "Error: %s\n", rt_last_error());
        return 1;
    }

    if (batch <= 0) batch = spp < 16 ? spp : (spp + 15) / 16;
    unsigned long long rays = 0;
    int done = 0;
    // a saved state belongs to one render: same scene, shader, samplers, light samples and depth (frame, samples, seed are in the header)
    const uint32_t tag = rth_render_tag((scene + "|" + shader + "|" + std::to_string(pixel_sampler) + "|" + std::to_string(light_sampler) + "|" +
                                         std::to_string(p.light_samples) + "|" + std::to_string(p.max_depth)).c_str());
    if (!resume.empty()) {
        if (FILE *probe = std::fopen(resume.c_str(), "rb")) {
            std::fclose(probe);
            if (rth_accum_load_tagged(resume.c_str(), width, height, spp, seed, tag, &done, accum.data()) != RT_OK) {
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
        if (!resume.empty() && rth_accum_save_tagged(resume.c_str(), width, height, spp, p.spp_end, seed, tag, accum.data()) != RT_OK) {
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
