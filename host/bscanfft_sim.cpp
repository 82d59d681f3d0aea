// bscanfft_sim.cpp -- headless counterpart of the reference's simulation harness
// (BscanFFTsim.cpp: "simulation using saved files, for testing and validation").
//
// The reference's loop (sim:775-1131) reads imgi.png each iteration, runs the OpenCV processing block
// (sim:842-955) and shows the B-scan; 'b' loads backg.png as data_yb (sim:803-813).  This program does the
// same through the C ABI of include/fdoct.h on an MI355X, from raw frame files (no OpenCV, no GUI):
//
//   bscanfft_sim --frames imgi_u16_96x128.bin --background backg_u16_96x128.bin
//                --width 128 --height 96 --bits 16 --numfftpoints 1024 --numdisplaypoints 512
//                [--averages A] [--sim] [--lambdamin 816e-9 --lambdamax 884e-9]
//                [--rowwisenormalize 0|1] [--donotnormalize 0|1] [--repeat K] [--threshold dB] --out prefix
//                [--gpus N [--devices d0,d1,...]] [--precise-division | --one-word-division]
//
// --gpus N: one process, N handles (fdoct_clone_to_device), one host thread per handle; the frames are sharded with
// fdoct_shard_frames (contiguous ranges, averaging groups never split -- the rule of the multi-process path,
// fdoct_amd/dist.py) and every handle writes its B-scans straight into its slice of the output.  No collective: the only
// exchange is the clone of the constant state.  --devices lists the device of each handle (default 0, 1, ..., wrapping
// around when fewer GPUs are visible -- several handles then share a device, which is how a 1-GPU box rehearses it).
//
// --frames holds one or more H x W frames back to back (u8 for --bits 8, little-endian u16 for --bits 16).
// Outputs: <prefix>_bscan.f32 / <prefix>_bscandb.f32 (reference layout D x H per B-scan, main:1220) and
// <prefix>.m with `bscan001=[...];` in the Matlab text form the reference's savematasdata writes
// (main:333-339) for the first B-scan, plus the display images the reference shows/saves (main:1242-1255, 1284,
// savematasimage): <prefix>_bscan001.pgm (grey) and <prefix>_bscanc001.ppm (colour-mapped).  Prints A-scans/s like the reference prints fps (sim:827-838).
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "../include/fdoct.h"
#include "ocv_io.h"

static bool ends_with(const std::string& s, const std::string& suf) {
  return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

static std::vector<unsigned char> read_file(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) {
    std::fprintf(stderr, "cannot open %s\n", path.c_str());
    std::exit(1);
  }
  return std::vector<unsigned char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
  std::string frames_path, bg_path, out = "bscanfft_sim";
  fdoct_config cfg;
  std::memset(&cfg, 0, sizeof cfg);
  cfg.struct_size = sizeof cfg;
  cfg.increasefftpointsmultiplier = 1;
  cfg.averages = 1;
  cfg.donotnormalize = 1;  // build/BscanFFT.ini default
  cfg.dc_mask = 1;
  cfg.lambdamin = 816e-9;  // sim:276-277
  cfg.lambdamax = 884e-9;
  int bits = 16, repeat = 1, gpus = 1;
  int precise = -1;  // -1: the library default (both words of 1/background since round 5)
  std::vector<int> devices;
  double bscanthreshold = -30.0;  // main:385
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    auto next = [&]() -> const char* {
      if (i + 1 >= argc) {
        std::fprintf(stderr, "missing value after %s\n", a.c_str());
        std::exit(1);
      }
      return argv[++i];
    };
    if (a == "--frames") frames_path = next();
    else if (a == "--background") bg_path = next();
    else if (a == "--out") out = next();
    else if (a == "--width") cfg.width = std::atoi(next());
    else if (a == "--height") cfg.height = std::atoi(next());
    else if (a == "--bits") bits = std::atoi(next());
    else if (a == "--numfftpoints") cfg.numfftpoints = std::atoi(next());
    else if (a == "--numdisplaypoints") cfg.numdisplaypoints = std::atoi(next());
    else if (a == "--averages") cfg.averages = std::atoi(next());
    else if (a == "--rowwisenormalize") cfg.rowwisenormalize = std::atoi(next());
    else if (a == "--donotnormalize") cfg.donotnormalize = std::atoi(next());
    else if (a == "--lambdamin") cfg.lambdamin = std::atof(next());
    else if (a == "--lambdamax") cfg.lambdamax = std::atof(next());
    else if (a == "--repeat") repeat = std::atoi(next());
    else if (a == "--threshold") bscanthreshold = std::atof(next());
    else if (a == "--sim") cfg.variant = FDOCT_VARIANT_SIM;
    else if (a == "--gpus") gpus = std::atoi(next());
    else if (a == "--precise-division") precise = 1;   // main:1132 divides in double: both words of 1/background on the fast path too (the default)
    else if (a == "--one-word-division") precise = 0;  // the opt-out: one f32 reciprocal on the fast path, for fringes above ~1 % of the DC level
    else if (a == "--devices") {
      for (const char* p = next(); *p;) {
        devices.push_back(std::atoi(p));
        while (*p && *p != ',') p++;
        if (*p == ',') p++;
      }
    }
    else {
      std::fprintf(stderr, "unknown option %s\n", a.c_str());
      return 1;
    }
  }
  if (frames_path.empty() || bg_path.empty() || cfg.width <= 0 || cfg.height <= 0 || cfg.numfftpoints <= 0) {
    std::fprintf(stderr, "usage: see the header of host/bscanfft_sim.cpp\n");
    return 1;
  }
  if (cfg.numdisplaypoints <= 0) cfg.numdisplaypoints = cfg.numfftpoints / 2;
  const fdoct_dtype dt = bits == 8 ? FDOCT_U8 : FDOCT_U16;
  const size_t es = bits == 8 ? 1 : 2;
  const size_t frame_bytes = (size_t)cfg.width * cfg.height * es;

  // frames saved by the instrument programs as .ocv Mat dumps (BscanFFTspinj.cpp:672-738) carry their own
  // geometry; raw .bin files take it from the command line
  std::vector<unsigned char> frames, bg;
  auto load = [&](const std::string& path, std::vector<unsigned char>* out) {
    if (ends_with(path, ".ocv")) {
      OcvMat m;
      if (!ocv_read(path, &m) || (m.depth != 0 && m.depth != 2) || m.channels != 1 || m.cols != cfg.width) {
        std::fprintf(stderr, "%s: not a single-channel 8/16-bit .ocv frame of width %d\n", path.c_str(), cfg.width);
        std::exit(1);
      }
      if ((m.depth == 0 ? 8 : 16) != bits) {
        std::fprintf(stderr, "%s holds %d-bit samples, --bits says %d\n", path.c_str(), m.depth == 0 ? 8 : 16, bits);
        std::exit(1);
      }
      *out = m.data;
    } else {
      *out = read_file(path);
    }
  };
  load(frames_path, &frames);
  load(bg_path, &bg);
  const int nframes_file = (int)(frames.size() / frame_bytes);
  if (nframes_file < 1) {
    std::fprintf(stderr, "%s holds no complete %dx%d frame\n", frames_path.c_str(), cfg.width, cfg.height);
    return 1;
  }
  const int nframes = nframes_file / cfg.averages * cfg.averages;
  if (nframes < 1) {
    std::fprintf(stderr, "need at least `averages` frames\n");
    return 1;
  }
  int bg_rows = 0;
  if (bg.size() >= frame_bytes) bg_rows = cfg.height;
  else if (bg.size() >= (size_t)cfg.width * es) bg_rows = 1;
  else {
    std::fprintf(stderr, "background file too small\n");
    return 1;
  }

  fdoct_handle h = nullptr;
  int rc = fdoct_create(&cfg, &h);  // replaces the one-time set-up sim:385-534, 765-773
  if (rc) {
    std::fprintf(stderr, "fdoct_create: %d %s\n", rc, fdoct_last_error(nullptr));
    return 1;
  }
  // the 'b' key: data_yb <- backg (sim:803-813)
  rc = fdoct_set_background(h, bg.data(), dt, bg_rows, 0);
  if (rc) {
    std::fprintf(stderr, "fdoct_set_background: %s\n", fdoct_last_error(h));
    return 1;
  }
  if (precise >= 0 && (rc = fdoct_set_precise_division(h, precise))) {
    std::fprintf(stderr, "fdoct_set_precise_division: %s\n", fdoct_last_error(h));
    return 1;
  }
  // before the loop: tables, kernel family and any run-time compile, so that the first frame does not stall (an acquisition
  // program would do the same after its 'b' key)
  const int prepared = fdoct_prepare(h, dt, FDOCT_LAYOUT_TRANSPOSED_DxH);
  if (prepared < 0) {
    std::fprintf(stderr, "fdoct_prepare: %d %s\n", prepared, fdoct_last_error(h));
    return 1;
  }
  const int G = nframes / cfg.averages;
  const size_t out_elems = (size_t)G * cfg.numdisplaypoints * cfg.height;
  std::vector<float> bscan(out_elems), bscandb(out_elems);
  // one handle per GPU: clones of the configured handle, each on its own device and host thread
  if (gpus < 1) gpus = 1;
  const int ndev = fdoct_device_count();
  std::vector<fdoct_handle> hs(gpus, nullptr);
  hs[0] = h;
  for (int g = 1; g < gpus; g++) {
    const int dev = g < (int)devices.size() ? devices[g] : (ndev > 0 ? g % ndev : 0);
    rc = fdoct_clone_to_device(h, dev, &hs[g]);
    if (rc) {
      std::fprintf(stderr, "fdoct_clone_to_device(%d): %d %s\n", dev, rc, fdoct_last_error(h));
      return 1;
    }
  }
  std::vector<int> rcs(gpus, 0);
  const size_t bscan_elems = (size_t)cfg.numdisplaypoints * cfg.height;
  auto run_shard = [&](int g) {
    int first = 0, count = 0;
    fdoct_shard_frames(nframes, cfg.averages, g, gpus, &first, &count);
    if (count == 0) return;
    const size_t o0 = (size_t)(first / cfg.averages) * bscan_elems;
    for (int k = 0; k < repeat && !rcs[g]; k++)  // the while(1) loop, bounded; sim:842-955 per iteration
      rcs[g] = fdoct_process(hs[g], frames.data() + (size_t)first * frame_bytes, dt, FDOCT_MEM_HOST, count, 0, bscan.data() + o0,
                             bscandb.data() + o0, FDOCT_MEM_HOST, FDOCT_LAYOUT_TRANSPOSED_DxH);
  };
  const auto t0 = std::chrono::steady_clock::now();
  if (gpus == 1) {
    run_shard(0);
  } else {
    std::vector<std::thread> th;
    for (int g = 0; g < gpus; g++) th.emplace_back(run_shard, g);
    for (auto& t : th) t.join();
  }
  for (int g = 0; g < gpus; g++)
    if (rcs[g]) {
      std::fprintf(stderr, "fdoct_process (handle %d): %d %s\n", g, rcs[g], fdoct_last_error(hs[g]));
      return 1;
    }
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  fdoct_timing tm;
  fdoct_get_timing(h, &tm);
  static const char* const family[] = {"none", "fused", "fused, D x H written by the chain", "fused, two stages", "wave per row",
                                       "wave per row, compiled for this geometry at run time", "workgroup per row", "long rows"};
  const int fam = fdoct_last_kernel(h);
  std::printf("%s: %d frame(s) x %d on %d handle(s), %d B-scan(s) %dx%d; %.0f A-scans/s incl. PCIe (device %.3f ms per call, kernel %.3f ms; %s kernel%s%s)\n",
              fdoct_version(), nframes, repeat, gpus, G, cfg.numdisplaypoints, cfg.height,
              (double)nframes * cfg.height * repeat / sec, tm.last_process_ms, tm.last_kernel_ms,
              fam >= 0 && fam < 8 ? family[fam] : "?", *fdoct_jit_note(h) ? "; " : "", fdoct_jit_note(h));

  {
    std::ofstream f(out + "_bscan.f32", std::ios::binary);
    f.write(reinterpret_cast<const char*>(bscan.data()), bscan.size() * sizeof(float));
    std::ofstream g(out + "_bscandb.f32", std::ios::binary);
    g.write(reinterpret_cast<const char*>(bscandb.data()), bscandb.size() * sizeof(float));
    // and the first B-scan as an .ocv Mat dump (CV_32F, D x H), the format savematasbin uses
    ocv_write(out + "_bscan001.ocv", cfg.numdisplaypoints, cfg.height, 5, bscan.data());
  }
  {
    // Matlab text, as savematasdata writes it (main:333-339: name "=" operator<<(Mat) ";"): rows separated by ";\n ",
    // columns by ", ", and every value with the 16 significant digits cv's default formatter gives a CV_64F Mat
    // (bscan is CV_64F in the reference, main:1220) -- enough to read the f32 results back exactly
    std::ofstream m(out + ".m");
    m << "bscan001=[";
    char num[40];
    for (int d = 0; d < cfg.numdisplaypoints; d++) {
      for (int r = 0; r < cfg.height; r++) {
        std::snprintf(num, sizeof num, "%.16g", (double)bscan[(size_t)d * cfg.height + r]);
        m << num;
        if (r + 1 < cfg.height) m << ", ";
      }
      if (d + 1 < cfg.numdisplaypoints) m << ";\n ";
    }
    m << "];\n";
  }
  {
    // the display chain of main:1242-1255 + 1284 for the first B-scan, as portable grey/pix maps
    const size_t px = (size_t)cfg.numdisplaypoints * cfg.height;
    std::vector<unsigned char> gray(px), bgr(3 * px);
    rc = fdoct_display(h, bscandb.data(), FDOCT_MEM_HOST, 1, cfg.numdisplaypoints, cfg.height, bscanthreshold, 0, gray.data(),
                       bgr.data(), FDOCT_MEM_HOST);
    if (rc) {
      std::fprintf(stderr, "fdoct_display: %d %s\n", rc, fdoct_last_error(h));
      return 1;
    }
    std::ofstream pg(out + "_bscan001.pgm", std::ios::binary);
    pg << "P5\n" << cfg.height << " " << cfg.numdisplaypoints << "\n255\n";
    pg.write(reinterpret_cast<const char*>(gray.data()), px);
    std::ofstream pp(out + "_bscanc001.ppm", std::ios::binary);
    pp << "P6\n" << cfg.height << " " << cfg.numdisplaypoints << "\n255\n";
    for (size_t i = 0; i < px; i++) {  // cv::Mat colour order is B,G,R; PPM wants R,G,B
      const char rgb[3] = {(char)bgr[3 * i + 2], (char)bgr[3 * i + 1], (char)bgr[3 * i]};
      pp.write(rgb, 3);
    }
  }
  for (fdoct_handle x : hs) fdoct_destroy(x);
  return 0;
}
