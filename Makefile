# Builds, in-tree:
#   raytracer-rs_amd/lib/librt_hip.so   device library (HIP kernels for gfx950 + the C ABI of include/rt_api.h)
#   raytracer-rs_amd/lib/librt_host.so  host scene library (include/rt_host.h)
#   tools/rtcli                         command-line renderer, counterpart of the reference's src/bin/main.rs
#   tools/ta_rate                       micro-benchmark of per-lane load rates (roofline of the traversal kernels)
#   oracle/build/liboracle_rt.so        CPU oracle (test infrastructure only)
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
ARCH ?= gfx950
PKG := raytracer-rs_amd
LIBDIR := $(PKG)/lib
CSRC := $(PKG)/csrc

# -ffp-contract=off: geometry must round exactly like a scalar CPU evaluation (no FMA fusion).
HIPFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -ffp-contract=off -fno-fast-math -Wall -Wno-unused-parameter
HOSTFLAGS := -O2 -std=c++17 -fPIC -Wall -Wextra

all: $(LIBDIR)/librt_hip.so $(LIBDIR)/librt_host.so tools/rtcli tools/ta_rate oracle

OBJDIR := $(LIBDIR)/obj
HIP_OBJS := $(OBJDIR)/rt_kernels.o $(OBJDIR)/rt_api.o $(OBJDIR)/rt_bvh.o $(OBJDIR)/rt_bvh_device.o
HIP_HDRS := $(CSRC)/rt_device.h $(CSRC)/rt_kernels.h $(CSRC)/rt_bvh.h $(CSRC)/rt_bvh_device.h include/rt_api.h

$(OBJDIR)/%.o: $(CSRC)/%.hip $(HIP_HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(OBJDIR)/%.o: $(CSRC)/%.cpp $(HIP_HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -x hip -c -o $@ $<

$(LIBDIR)/librt_hip.so: $(HIP_OBJS)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $(HIP_OBJS)

$(LIBDIR)/librt_host.so: $(PKG)/host/rt_host.cpp $(PKG)/host/rt_host.hpp include/rt_host.h include/rt_api.h
	@mkdir -p $(LIBDIR)
	$(CXX) $(HOSTFLAGS) -shared -o $@ $(PKG)/host/rt_host.cpp

tools/rtcli: tools/rtcli.cpp $(LIBDIR)/librt_hip.so $(LIBDIR)/librt_host.so
	$(CXX) $(HOSTFLAGS) -pthread -o $@ tools/rtcli.cpp -L$(LIBDIR) -lrt_hip -lrt_host -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)'

# micro-benchmark: peak rate of per-lane (divergent) loads, the roofline of the BVH walk kernels
tools/ta_rate: tools/ta_rate.hip
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -o $@ $<

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(LIBDIR) tools/rtcli tools/ta_rate oracle/build

.PHONY: all oracle clean
