# Builds libportal_amd.so (C ABI, include/portal_amd.h), the CLI and the fixed gfx950 helpers.
#   make            -> portal_amd/libportal_amd.so + portal_amd/portal-amd
#   make kernels    -> portal_amd/kernels/*.hsaco (hipcc --genco, gfx950)
CXX      ?= g++
HIPCC    ?= /opt/rocm/bin/hipcc
CXXFLAGS ?= -O2 -g -std=c++20 -fPIC -Wall -Wextra -Wno-unused-parameter
HOST     := portal_amd/csrc/host
DEVICE   := portal_amd/csrc/device
OBJDIR   := build/obj
SRCS     := ron.cpp formula.cpp scene.cpp glsl_translate.cpp glsl_bound.cpp glsl_hoist.cpp codegen.cpp embedded.cpp hip_api.cpp kernel.cpp postprocess.cpp png_io.cpp capi.cpp multigpu.cpp
OBJS     := $(SRCS:%.cpp=$(OBJDIR)/%.o)
LIB      := portal_amd/libportal_amd.so
CLI      := portal_amd/portal-amd
KERNELS  := portal_amd/kernels/fb_store.hsaco portal_amd/kernels/average_images.hsaco

all: $(LIB) $(CLI)

$(HOST)/embedded_device_sources.inc: $(DEVICE)/ptl_glsl.h $(DEVICE)/ptl_library.h $(DEVICE)/ptl_trace.tpl $(DEVICE)/ptl_entry.h portal_amd/csrc/embed_sources.py
	python3 portal_amd/csrc/embed_sources.py $@ device_source_glsl=$(DEVICE)/ptl_glsl.h device_source_library=$(DEVICE)/ptl_library.h \
	    device_source_trace_template=$(DEVICE)/ptl_trace.tpl device_source_entry=$(DEVICE)/ptl_entry.h

$(OBJDIR)/embedded.o: $(HOST)/embedded_device_sources.inc
$(OBJDIR)/%.o: $(HOST)/%.cpp $(wildcard $(HOST)/*.h) include/portal_amd.h
	@mkdir -p $(OBJDIR)
	$(CXX) $(CXXFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(CXX) -shared -o $@ $(OBJS) -ldl -lz -lpthread

$(CLI): $(HOST)/cli.cpp $(LIB) include/portal_amd.h
	$(CXX) $(CXXFLAGS) $(HOST)/cli.cpp -o $@ -Lportal_amd -lportal_amd -Wl,-rpath,'$$ORIGIN'

kernels: $(KERNELS)
portal_amd/kernels/%.hsaco: portal_amd/csrc/kernels/%.hip
	@mkdir -p portal_amd/kernels
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -mllvm -vgpr-regalloc=basic --genco --no-gpu-bundle-output $< -o $@  # allocator: see kernel.cpp

clean:
	rm -rf build $(LIB) $(CLI) portal_amd/kernels $(HOST)/embedded_device_sources.inc

.PHONY: all kernels clean
