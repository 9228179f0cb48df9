/* Plain-C client of the ABI in include/vqvs.h (no Python, no torch): loads libvqvs_hip.so, prints the library version
 * and the parameter table the library expects for a UNetPredictor of the given width -- the first thing a host in any
 * language does before feeding it a {"kwargs","state_dict"} checkpoint (reference vq_voice_swap/models/base.py:74-104).
 *
 *   gcc -std=c99 -Iinclude examples/abi_info.c -ldl -o /tmp/abi_info && /tmp/abi_info vq_voice_swap_amd/libvqvs_hip.so 64
 *
 * Only host-side entry points are called, so it also runs on a machine without a GPU. */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vqvs.h"

typedef const char* (*version_fn)(void);
typedef int (*count_fn)(const vqvs_cfg*);
typedef int (*info_fn)(const vqvs_cfg*, int, char*, int, int64_t*, int*);

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s <path to libvqvs_hip.so> [base_channels]\n", argv[0]);
    return 2;
  }
  void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!lib) {
    fprintf(stderr, "dlopen failed: %s\n", dlerror());
    return 1;
  }
  version_fn version = (version_fn)dlsym(lib, "vqvs_version");
  count_fn count = (count_fn)dlsym(lib, "vqvs_param_count");
  info_fn info = (info_fn)dlsym(lib, "vqvs_param_info");
  if (!version || !count || !info) {
    fprintf(stderr, "missing symbols\n");
    return 1;
  }
  vqvs_cfg cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.kind = VQVS_KIND_PREDICTOR;
  cfg.base_channels = argc > 2 ? atoi(argv[2]) : 64;
  cfg.in_channels = 1;
  cfg.out_channels = 1;
  cfg.precision = VQVS_PREC_F16;
  cfg.max_batch = 1;
  cfg.max_T = 256;
  const int n = count(&cfg);
  if (n < 0) {
    fprintf(stderr, "vqvs_param_count failed (%d)\n", n);
    return 1;
  }
  printf("%s: %d parameters for UNetPredictor(base_channels=%d)\n", version(), n, cfg.base_channels);
  long long total = 0;
  for (int i = 0; i < n; ++i) {
    char name[256];
    int64_t shape[4];
    int nd = 0;
    if (info(&cfg, i, name, (int)sizeof(name), shape, &nd) != 0) return 1;
    long long numel = 1;
    for (int k = 0; k < nd; ++k) numel *= shape[k];
    total += numel;
    if (i < 4 || i == n - 1) printf("  %-44s [%lld x %lld x %lld]\n", name, (long long)shape[0], (long long)shape[1], (long long)shape[2]);
    if (i == 4) printf("  ...\n");
  }
  printf("total %lld float32 values\n", total);
  dlclose(lib);
  return 0;
}
