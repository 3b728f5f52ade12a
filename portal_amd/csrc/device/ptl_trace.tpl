// ptl_trace.tpl -- per-scene kernel template, filled in by codegen.cpp (apply_template).
//
// Plays the role of the reference's src/frag.glsl: the slash-slash-percent markers are the slots
// Scene::generate_shader_code fills (src/gui/scene.rs:693-1110, src/code_generation.rs:81-98).
// The text around the slots is a from-scratch HIP C++ restatement of the per-pixel path
// (AA loop -> primary ray -> bounce loop -> shade), laid out for gfx950: one 64-lane
// wavefront owns an 8x8 pixel tile, four tiles form a 32x8 block, uniforms live in one
// __constant__ block read through the scalar cache, the RGBA8 tile is transposed through
// LDS so that every wave stores two full 128-byte rows.
//%predefined_library//%

// (the bounce-loop trip counter PTL_COUNT_SEGMENT() -- and the checking build's PTL_NOTE_NOT_AFFINE() -- are declared at the top of ptl_glsl.h)

namespace glsl {

// --- scene uniforms (reference: `uniform ...;` declarations, scene.rs:661-718) -------------
//%uniforms//%

// --- material ids (scene.rs:720-845) ---------------------------------------------------------
//%materials_defines//%

// Everything that reads scene uniforms lives in one struct whose only state is a pointer to the
// uniform block; the uniform accessors (`a_mat`, `progress_u`, `_camera` ...) expand to loads through
// that pointer.  The bounce loop re-launders the pointer once per trip (PTL_RELAUNDER), so the
// scalar loads of the portal matrices can be shared inside a trip but cannot be hoisted above the loop,
// where hipcc would otherwise keep hundreds of SGPRs live and spill them to VGPR lanes
// (measured: triple_portal 4K 3.36 ms -> 1.2 ms, profiles/r01/variants5_uniform_reload.jsonl).
struct ptl_tracer {
    const ptl_uniform_block* ptl_ubp;
#if !(PTL_DEVICE_BUILD && (defined(PTL_UNIFORMS_IN_LDS) || defined(PTL_UNIFORM_RELOAD) || defined(PTL_UNIFORM_HOIST)))
#undef PTL_U
#define PTL_U (*ptl_ubp)
#endif
#if PTL_DEVICE_BUILD && !defined(PTL_UNIFORMS_IN_LDS) && !defined(PTL_UNIFORM_RELOAD) && !defined(PTL_UNIFORM_HOIST)
#if !defined(PTL_LAUNDER_POINTER)
// What is laundered is an OFFSET (an opaque zero in an SGPR), not the pointer: the address stays visibly inside the __constant__
// object, so the loads keep its address space and are SCALAR loads (s_load into SGPRs, operands of the VALU instructions as they
// are).  Through a laundered POINTER (-DPTL_LAUNDER_POINTER, the round-1 form) the compiler only sees a generic address and reads
// every uniform with a vector load -- one VGPR per dword, 185 VGPRs and two waves per SIMD for the un-specialised portal_in_portal
// kernel: 1.30 ms against 0.84 ms (profiles/r02/variants7_scalar_uniform_loads.jsonl; same frames).
#define PTL_RELAUNDER()                                   \
    do {                                                  \
        int ptl_zero = 0;                                 \
        asm volatile("" : "+s"(ptl_zero));                \
        ptl_ubp = reinterpret_cast<const ptl_uniform_block*>(reinterpret_cast<const char*>(&ptl_u) + ptl_zero); \
    } while (0)
#else
#define PTL_RELAUNDER()                                   \
    do {                                                  \
        const ptl_uniform_block* ptl_fresh = &ptl_u;      \
        asm volatile("" : "+s"(ptl_fresh)); /* opaque, uniform (SGPR) */ \
        ptl_ubp = ptl_fresh;                              \
    } while (0)
#endif
#else
#define PTL_RELAUNDER() ((void)0)
#endif

#if defined(PTL_NO_PIXEL_RELAUNDER)
#define PTL_PIXEL_RELAUNDER() ((void)0)
#else
#define PTL_PIXEL_RELAUNDER() PTL_RELAUNDER()
#endif

// Scene snippets are plain GLSL functions without HIP attributes: let clang treat every
// function declared in this region as __host__ __device__.
#if PTL_DEVICE_BUILD
#pragma clang force_cuda_host_device begin
#endif

// --- scene library snippets (scene.rs:1037-1044) ---------------------------------------------
//%library//%

// --- is_inside_N / intersect_N wrappers around scene snippets (scene.rs:847-883) ------------
//%intersection_functions//%

// --- intersect_material_N (scene.rs:1011-1024) -----------------------------------------------
//%intersection_material_functions//%

#if PTL_DEVICE_BUILD
#pragma clang force_cuda_host_device end
#endif

// Nearest object hit along r: one generated statement group per scene object.  (reference shell: src/frag.glsl:19-31)
// `ptl_far`: a distance beyond which a hit cannot matter to the caller (the bounce loop passes the hit distance of the scene's
// intersection-material snippet, which it evaluates first: a plane farther than that loses against the snippet whatever it is).  It
// only tightens the bound of the wave-level plane cull (ptl_plane_cull): planes that are tested are tested in full.
PTL_FN SceneIntersection scene_intersect(const Ray& r, float ptl_far = __builtin_inff()) {
    SceneIntersection i = SceneIntersection{0, intersection_none, false};
    SceneIntersection ihit = SceneIntersection{0, intersection_none, false};
    SurfaceIntersection hit = intersection_none;
    vec3 normal = vec3(0.0f);
    int inside = NOT_INSIDE;
    float len = 1.0f;
    Ray transformed_ray = ray_none;
    bool flipped = false;  // plane tests with a derived entry: the unit normal was turned to face the ray
    (void)ihit; (void)hit; (void)normal; (void)inside; (void)len; (void)transformed_ray; (void)flipped;

//%intersections//%

    return i;
}

#ifdef PTL_FIRST_TRIP_PLANES
// The same function for the trip on which every ray of the wave still starts at the camera (KernelOptions::first_trip_planes): the
// generated plane tests take `plane_inv * r.o` -- one value for the whole frame -- from the prologue kernel's `ptl_dvo_<object>_<side>`
// (derive() below: the same product of the same matrix and the same origin), everything else is the statement list above.
PTL_FN SceneIntersection scene_intersect_first(const Ray& r, float ptl_far = __builtin_inff()) {
    SceneIntersection i = SceneIntersection{0, intersection_none, false};
    SceneIntersection ihit = SceneIntersection{0, intersection_none, false};
    SurfaceIntersection hit = intersection_none;
    vec3 normal = vec3(0.0f);
    int inside = NOT_INSIDE;
    float len = 1.0f;
    Ray transformed_ray = ray_none;
    bool flipped = false;
    (void)ihit; (void)hit; (void)normal; (void)inside; (void)len; (void)transformed_ray; (void)flipped;

//%intersections_first//%

    return i;
}
#endif

// Prologue (ptl_derive_kernel, once per uniform upload): the ray-independent part of the plane tests above whose matrices
// are run-time uniforms -- the unit normal plane_intersect would normalise on every call and the two verdicts
// is_collinear(hit.n, normal) can have (hit.n is that unit normal or its negation).  Same functions, same operations as the
// plain form, evaluated once per frame instead of once per trip and lane; the results land behind the uploaded uniforms.
#define PTL_DV_OUT (*out)
PTL_FN void derive(ptl_uniform_block* out) {
    (void)out;
#ifdef PTL_DERIVED_BUILTINS
    // per-PIXEL work that depends on nothing but the frame's builtins: the ray origins, tan(fov / 2), the reciprocal of the frame size
    // -- ~100 VALU instructions every pixel would otherwise repeat (tan alone is two polynomial kernels and a division)
    out->ptl_dv_origin = ptl_mul_runtime(_camera, vec4(0.0f, 0.0f, 0.0f, 1.0f));
    out->ptl_dv_origin_left = _camera_left_eye * vec4(0.0f, 0.0f, 0.0f, 1.0f);
    out->ptl_dv_origin_right = _camera_right_eye * vec4(0.0f, 0.0f, 0.0f, 1.0f);
    out->ptl_dv_tan_half_view = tan(ptl_div(_view_angle, 2.0f));
    out->ptl_dv_pixel_size = ptl_rcp(min(_resolution.x, _resolution.y));
    out->ptl_dv_half_resolution = _resolution / 2.0f;
#endif
//%derive//%
}

// Material id -> what happens to the path. (reference shell: src/frag.glsl:33-50)
PTL_FN MaterialProcessing material_process(Ray r, const SceneIntersection& i) {
    SurfaceIntersection hit = i.hit;
    if (i.in_subspace) r.in_subspace = !r.in_subspace;
    if (i.material == 0) {
#ifdef PTL_MATERIAL_TABLE
    // every Simple material of the scene and the three DEBUG_* ones: their nine literals from the LDS table, ONE copy of the body (codegen.cpp, materials)
    } else if (ptl_material_in_table(i.material)) {
        struct alignas(16) ptl_words4 { unsigned int x, y, z, w; };
#if PTL_DEVICE_BUILD && PTL_MATERIAL_TABLE == 2
        // the table in constant memory, one scalar load per DISTINCT material of the wave (usually one): the nine values arrive in SGPRs, the grid flags
        // branch on the scalar unit, and the body runs once per material present
        MaterialProcessing shaded = material_empty();
        for (bool done = false; !done;) {
            const int current = __builtin_amdgcn_readfirstlane(i.material);
            // (the index through an opaque SGPR: where the compiler knows `current == i.material` it indexes with the per-lane value -- vector loads)
            int row = 8 * current;
            asm volatile("" : "+s"(row));
            const ptl_words4 a = *reinterpret_cast<const ptl_words4*>(&ptl_material_table[row]);
            const ptl_words4 b = *reinterpret_cast<const ptl_words4*>(&ptl_material_table[row + 4]);
            if (i.material == current) {
                shaded = material_simple2(hit, r, vec3(__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, a.y), __builtin_bit_cast(float, a.z)), __builtin_bit_cast(float, a.w),
                                          (b.z & 1u) != 0, __builtin_bit_cast(float, b.x), __builtin_bit_cast(float, b.y), (b.z & 2u) != 0, (b.z & 4u) != 0);
                done = true;
            }
        }
        return shaded;
#else
        const ptl_words4 a = *reinterpret_cast<const ptl_words4*>(&ptl_material_table[8 * i.material]);
        const ptl_words4 b = *reinterpret_cast<const ptl_words4*>(&ptl_material_table[8 * i.material + 4]);
        return material_simple2(hit, r, vec3(__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, a.y), __builtin_bit_cast(float, a.z)), __builtin_bit_cast(float, a.w),
                                (b.z & 1u) != 0, __builtin_bit_cast(float, b.x), __builtin_bit_cast(float, b.y), (b.z & 2u) != 0, (b.z & 4u) != 0);
#endif
#else
    } else if (i.material == DEBUG_RED) {
        return material_simple2(hit, r, color(0.9f, 0.2f, 0.2f), 0.5f, false, 1.0f, 0.0f, false, false);
    } else if (i.material == DEBUG_GREEN) {
        return material_simple2(hit, r, color(0.2f, 0.9f, 0.2f), 0.5f, false, 1.0f, 0.0f, false, false);
    } else if (i.material == DEBUG_BLUE) {
        return material_simple2(hit, r, color(0.2f, 0.2f, 0.9f), 0.5f, false, 1.0f, 0.0f, false, false);
#endif

//%material_processing//%

    }
    return material_final(vec3(0.0f));  // unknown material id
}

// Scene snippets that return hit and material in one go. (src/frag.glsl:52-59)
// `ptl_far`: a distance beyond which a hit cannot matter to the caller (PTL_BOUNDED_SNIPPETS: the bounce loop passes the hit distance of
// scene_intersect(), which it then evaluates first; the snippets whose shape allows it skip candidates beyond it, glsl_translate.h).
PTL_FN SceneIntersectionWithMaterial scene_intersect_material_process(const Ray& r, float ptl_far = __builtin_inff()) {
    (void)ptl_far;
    SceneIntersectionWithMaterial result = SceneIntersectionWithMaterial{scene_intersection_none, material_empty()};
    SceneIntersectionWithMaterial hit = SceneIntersectionWithMaterial{scene_intersection_none, material_empty()};
    (void)hit;

//%intersection_material_processing//%

    return result;
}

#ifdef PTL_FIRST_TRIP_SNIPPETS
// First-trip variants (KernelOptions::first_trip, flags bit 13).  Every primary ray of the frame starts at the camera: on the first
// trip of the bounce loop -- nine trips in ten on the headline frame -- the ORIGIN half of whatever the scene's intersection-material
// snippets do to the ray depends on uniforms alone.  The code generator emits a second copy of each snippet, `intersect_material_<N>_first`,
// in which the hoister (host/glsl_hoist.h) knows that about `r`: origins of `transform(uniform matrix, ray)` chains come from the
// prologue kernel's tables, the direction half stays per ray.  Same operations on the same values; later trips use the general copy.
PTL_FN SceneIntersectionWithMaterial scene_intersect_material_process_first(const Ray& r, float ptl_far = __builtin_inff()) {
    (void)ptl_far;
    SceneIntersectionWithMaterial result = SceneIntersectionWithMaterial{scene_intersection_none, material_empty()};
    SceneIntersectionWithMaterial hit = SceneIntersectionWithMaterial{scene_intersection_none, material_empty()};
    (void)hit;

//%intersection_material_processing_first//%

    return result;
}
#endif

// --- the bounce loop -------------------------------------------------- src/frag.glsl:74-159
struct RayTraceResult {
    vec3 color;
    float depth;
    bool has_depth;
};

PTL_FN float normalize_depth_value(float depth) {  // frag.glsl:80-84
    float depth_min = min(_depth_map_min, _depth_map_max);
    float depth_max = max(_depth_map_min, _depth_map_max);
    return clamp(ptl_div(depth - depth_min, max(1e-6f, depth_max - depth_min)), 0.0f, 1.0f);
}

PTL_FN vec3 depth_gradient_inferno(float t) {  // frag.glsl:86-99
    vec3 c0 = sqrvec(vec3(0.001462f, 0.000466f, 0.013866f));
    vec3 c1 = sqrvec(vec3(0.258234f, 0.038571f, 0.406485f));
    vec3 c2 = sqrvec(vec3(0.578304f, 0.148039f, 0.404411f));
    vec3 c3 = sqrvec(vec3(0.865006f, 0.316822f, 0.226055f));
    vec3 c4 = sqrvec(vec3(0.987622f, 0.645320f, 0.039886f));
    vec3 c5 = sqrvec(vec3(0.988362f, 0.998364f, 0.644924f));
    if (t < 0.2f) return mix(c0, c1, ptl_div(t, 0.2f));
    if (t < 0.4f) return mix(c1, c2, ptl_div(t - 0.2f, 0.2f));
    if (t < 0.6f) return mix(c2, c3, ptl_div(t - 0.4f, 0.2f));
    if (t < 0.8f) return mix(c3, c4, ptl_div(t - 0.6f, 0.2f));
    return mix(c4, c5, ptl_div(t - 0.8f, 0.2f));
}

PTL_FN vec3 sample_depth_gradient(float depth) {  // frag.glsl:101-104
    return depth_gradient_inferno(1.0f - normalize_depth_value(depth));
}

// One pass of the bounce loop for one ray: nearest hit, material, advance.  Returns true when the path has ended (`out` is
// its result), false when `r` / `current_color` / `all_t` have been advanced to the next segment.  (frag.glsl:113-156)
#ifdef PTL_FIRST_TRIP
// (`first_form`, wave-uniform: r still starts at the camera, so the snippets run in their first-trip form -- see scene_intersect_material_process_first)
PTL_FN bool trace_segment(Ray& r, vec3& current_color, float& all_t, float camera_scale, const vec3& not_found_color, RayTraceResult& out, bool first_form) {
#else
PTL_FN bool trace_segment(Ray& r, vec3& current_color, float& all_t, float camera_scale, const vec3& not_found_color, RayTraceResult& out) {
#endif
    r = ptl_affine(r);  // PTL_AFFINE_RAYS: o.w = 1 and d.w = 0 spelled once per trip (ptl_library.h); otherwise nothing
#if defined(PTL_BOUNDED_SNIPPETS) && !defined(PTL_SNIPPETS_FIRST)
    // Like the reference (frag.glsl:114-115): scene_intersect first.  Its hit distance then bounds the snippets: a candidate of theirs
    // beyond it could never be the `nearer` one below, and the snippets the generator could prove it for skip such candidates
    // (KernelOptions::bound_snippets) -- for the headline scene that is most of what its snippet does.  (-DPTL_SNIPPETS_FIRST: A/B.)
#ifdef PTL_FIRST_TRIP_PLANES
    SceneIntersection i = first_form ? scene_intersect_first(r) : scene_intersect(r);
#else
    SceneIntersection i = scene_intersect(r);
#endif
    const float ptl_bound = i.hit.hit ? i.hit.t : __builtin_inff();
#ifdef PTL_FIRST_TRIP_SNIPPETS
    SceneIntersectionWithMaterial i2 = first_form ? scene_intersect_material_process_first(r, ptl_bound) : scene_intersect_material_process(r, ptl_bound);
#else
    SceneIntersectionWithMaterial i2 = scene_intersect_material_process(r, ptl_bound);
#endif
#else
    // The reference evaluates scene_intersect first (frag.glsl:114-115); both are pure, and with the snippet's hit distance known the
    // plane tests beyond it can be culled: `i` then is the nearest object in front of the snippet's hit, or whatever else survived --
    // and whenever the two differ the snippet's hit is nearer than both and is what gets used below.
#ifdef PTL_FIRST_TRIP_SNIPPETS
    SceneIntersectionWithMaterial i2 = first_form ? scene_intersect_material_process_first(r) : scene_intersect_material_process(r);
#else
    SceneIntersectionWithMaterial i2 = scene_intersect_material_process(r);
#endif
    const float ptl_bound = (i2.scene.hit.hit && i2.scene.hit.t > 0.0f) ? i2.scene.hit.t : __builtin_inff();
#ifdef PTL_FIRST_TRIP_PLANES
    SceneIntersection i = first_form ? scene_intersect_first(r, ptl_bound) : scene_intersect(r, ptl_bound);
#else
    SceneIntersection i = scene_intersect(r, ptl_bound);
#endif
#endif
#if defined(PTL_FIRST_TRIP) && !defined(PTL_FIRST_TRIP_SNIPPETS) && !defined(PTL_FIRST_TRIP_PLANES)
    (void)first_form;
#endif

    // `m` is left unset by the reference when a snippet reports hit with t <= 0 (GLSL:
    // undefined value); this build defines that case as the all-zero MaterialProcessing.
    MaterialProcessing m = MaterialProcessing{false, vec3(0.0f), ray_none};
    if (nearer(i.hit, i2.scene.hit)) {
        r.o += r.d * i2.scene.hit.t;
        all_t += i2.scene.hit.t * r.tmul;
        if (i2.scene.material == CUSTOM_MATERIAL) {
            m = i2.material;
        } else {
            m = material_process(r, i2.scene);
        }
    } else if (i.hit.hit) {
        r.o += r.d * i.hit.t;
        all_t += i.hit.t * r.tmul;
        m = material_process(r, i);
    }

    if (!(i.hit.hit || i2.scene.hit.hit)) {  // escaped the scene
        out = r.in_subspace ? RayTraceResult{color(0.0f, 0.0f, 0.0f), 0.0f, false} : RayTraceResult{current_color * not_found_color, 0.0f, false};
        return true;
    }
    current_color *= m.mul_to_color;
    if (m.is_final) {
        float depth = ptl_div(all_t, max(camera_scale, 1e-6f));
        if (all_t > _t_start * camera_scale && _darken_by_distance == 1) {  // fade to black with distance
            if (all_t > _t_end * camera_scale) all_t = _t_end * camera_scale;
            float gray_t = ptl_div(ptl_div(all_t - _t_start * camera_scale, _t_end - _t_start), camera_scale);
            out = RayTraceResult{color(0.0f, 0.0f, 0.0f) * sqr(sqr(gray_t)) + current_color * sqr(sqr(1.0f - gray_t)), depth, true};
            return true;
        }
        out = RayTraceResult{current_color, depth, true};
        return true;
    }
    r = m.new_ray;
    return false;
}


// The bounce loop.  A wavefront owns an 8x8 pixel tile (ptl_entry.h); its 64 rays take different numbers of trips.  On the
// device the loop is written for the wave: a lane whose path has ended drops out of `alive`, and the whole wave leaves as
// soon as a ballot over `alive` comes back empty -- one scalar compare per trip, no lane ever waits for a loop counter it
// no longer needs.  (-DPTL_NO_WAVE_LOOP: the per-lane form with returns, which the compiler turns into the same shape.)
#ifdef PTL_FIRST_TRIP
PTL_FN RayTraceResult ray_tracing(Ray r, float camera_scale, bool origin_is_camera) {  // origin_is_camera: r.o is PTL_U.ptl_dv_origin, bit for bit
#else
PTL_FN RayTraceResult ray_tracing(Ray r, float camera_scale) {
#endif
    //%skybox_processing//%

    vec3 current_color = vec3(1.0f);
    float all_t = 0.0f;
    RayTraceResult result = RayTraceResult{color(0.0f, 0.0f, 0.0f), 0.0f, false};  // depth exhausted
#if PTL_DEVICE_BUILD && !defined(PTL_NO_WAVE_LOOP)
    bool alive = true;
#ifdef PTL_FIRST_TRIP
    const bool first_form = __builtin_amdgcn_ballot_w64(!origin_is_camera) == 0ull;  // one decision per wave (side-by-side stereo mixes eyes)
#endif
#if defined(PTL_FIRST_TRIP) && defined(PTL_PEEL_FIRST_TRIP)
    // Trip 0 apart from the loop: it is the only trip that can take the first-trip forms, every lane is alive in it, and nine trips in
    // ten are a trip 0 -- with it peeled, the loop body has the general forms only and trip 0 the first-trip forms only (when the wave
    // agrees it starts at the camera), instead of one body that decides per trip.
    int j = 0;
    if (_ray_tracing_depth > 0) {
        PTL_RELAUNDER();
        PTL_COUNT_SEGMENT();
        alive = !trace_segment(r, current_color, all_t, camera_scale, not_found_color, result, first_form);
        j = 1;
    }
    for (; j < _ray_tracing_depth; j++) {
        if (__builtin_amdgcn_ballot_w64(alive) == 0ull) break;
        PTL_RELAUNDER();
        if (alive) {
            PTL_COUNT_SEGMENT();
            alive = !trace_segment(r, current_color, all_t, camera_scale, not_found_color, result, false);
        }
    }
#else
    for (int j = 0; j < _ray_tracing_depth; j++) {
        if (__builtin_amdgcn_ballot_w64(alive) == 0ull) break;  // every ray of the tile has terminated
        PTL_RELAUNDER();
        if (alive) {
            PTL_COUNT_SEGMENT();
#ifdef PTL_FIRST_TRIP
            alive = !trace_segment(r, current_color, all_t, camera_scale, not_found_color, result, j == 0 && first_form);
#else
            alive = !trace_segment(r, current_color, all_t, camera_scale, not_found_color, result);
#endif
        }
    }
#endif  // PTL_PEEL_FIRST_TRIP
#else
    for (int j = 0; j < _ray_tracing_depth; j++) {
        PTL_RELAUNDER();
        PTL_COUNT_SEGMENT();
#ifdef PTL_FIRST_TRIP
        if (trace_segment(r, current_color, all_t, camera_scale, not_found_color, result, j == 0 && origin_is_camera)) return result;
#else
        if (trace_segment(r, current_color, all_t, camera_scale, not_found_color, result)) return result;
#endif
    }
#endif
    return result;
}

// --- camera teleportation -------------------------------------------- src/frag.glsl:199-257
// Follows the segment a -> b (r.o = a, r.d = b - a, so the segment is t*tmul in [0, 1]) through up to
// ten portals and returns where b ends up.  The reference packs the three floats into RGBA8 pixels
// (encode_float, frag.glsl:166-197,527-548) because GL can only return colours; here the kernel
// returns the struct itself, `teleported` standing for the reference's "(x, y, z) != 0" test
// (src/main.rs:1400-1408).
struct ExternalRayTeleportation {
    vec3 pos;
    bool encounter_object;
    bool change_subspace;
    bool teleported;
};

PTL_FN ExternalRayTeleportation teleport_external_ray(Ray r) {
    r = normalize_ray(r);
    bool have_result = false;
    bool stop_at_object = false;
    float all_t = 0.0f;
    const int max_camera_teleports = 10;
    for (int j = 0; j < max_camera_teleports; j++) {
        PTL_RELAUNDER();
        SceneIntersection i = scene_intersect(r);
        SceneIntersectionWithMaterial i2 = scene_intersect_material_process(r);
        bool continue_intersect = false;
        MaterialProcessing m = MaterialProcessing{false, vec3(0.0f), ray_none};
        if (nearer(i.hit, i2.scene.hit)) {
            if (i2.scene.hit.t * r.tmul + all_t < 1.0f) {
                r.o += r.d * i2.scene.hit.t;
                all_t += i2.scene.hit.t * r.tmul;
                if (i2.scene.material == CUSTOM_MATERIAL) {
                    m = i2.material;
                } else {
                    m = material_process(r, i2.scene);
                }
                continue_intersect = !m.is_final;
                stop_at_object = stop_at_object || m.is_final;
            }
        } else if (i.hit.hit) {
            if (i.hit.t * r.tmul + all_t < 1.0f) {
                r.o += r.d * i.hit.t;
                all_t += i.hit.t * r.tmul;
                m = material_process(r, i);
                continue_intersect = !m.is_final;
                stop_at_object = stop_at_object || m.is_final;
            }
        }
        if (!continue_intersect) break;
        r = m.new_ray;
        have_result = true;
    }
    bool change_subspace = int(r.in_subspace) != _camera_in_subspace;
    if (have_result) {
        r.o += r.d * (1.0f - all_t) / r.tmul;
        vec3 p = r.o.sw<0, 1, 2>();
        return ExternalRayTeleportation{p, stop_at_object, change_subspace, !(p.x == 0.0f && p.y == 0.0f && p.z == 0.0f)};
    }
    return ExternalRayTeleportation{vec3(0.0f), stop_at_object, change_subspace, false};
}

// The launchable form: reads the segment from the _external_ray_a/_b uniforms (frag.glsl:529) and
// writes {x, y, z, teleported, encounter_object, change_subspace}.
PTL_FN void teleport_external_ray_entry(float* out6) {
    ExternalRayTeleportation t = teleport_external_ray(
        Ray{vec4(_external_ray_a, 1.0f), vec4(_external_ray_b - _external_ray_a, 0.0f), 1.0f, _camera_in_subspace == 1});
    out6[0] = t.pos.x;
    out6[1] = t.pos.y;
    out6[2] = t.pos.z;
    out6[3] = t.teleported ? 1.0f : 0.0f;
    out6[4] = t.encounter_object ? 1.0f : 0.0f;
    out6[5] = t.change_subspace ? 1.0f : 0.0f;
}

// --- camera ---------------------------------------------------------- src/frag.glsl:297-342
PTL_FN float Pow2(float x) { return x * x; }

// Pannini inverse mapping, tc in [-1,1]^2, fov in [0,pi), d in [0,1] (after shadertoy Wt3fzB)
PTL_FN vec3 PaniniProjection(vec2 tc, float fov, float d) {
    const float Pi05 = 3.14159265359f * 0.5f;
    float d2 = d * d;
    {
        float fo = Pi05 - fov * 0.5f;
        float f = ptl_div(cos(fo), sin(fo));
        float f2 = f * f;
        float b = ptl_div(sqrt(max(0.0f, Pow2(d + d2) * (f2 + f2 * f2))) - (d * f + f), d2 + d2 * f2 - 1.0f);
        tc *= b;
    }
    float h = tc.x;
    float v = tc.y;
    float h2 = h * h;
    float k = ptl_div(h2, Pow2(d + 1.0f));
    float k2 = k * k;
    float discr = max(0.0f, k2 * d2 - (k + 1.0f) * (k * d2 - 1.0f));
    float cosPhi = ptl_div(-k * d + sqrt(discr), k + 1.0f);
    float S = ptl_div(d + 1.0f, d + cosPhi);
    float tanTheta = ptl_div(v, S);
    float sinPhi = sqrt(max(0.0f, 1.0f - Pow2(cosPhi)));
    if (tc.x < 0.0f) sinPhi *= -1.0f;
    float s = inversesqrt(1.0f + Pow2(tanTheta));
    return vec3(sinPhi, tanTheta, cosPhi) * s;
}

// camera_matrix * v.  On the device the matrix is named, not passed: `which_eye` says which of the three uniform matrices it is (0 the
// camera, 1 / 2 the left / right eye).  Side-by-side stereo picks the eye by pixel column, so `which_eye` can differ between the lanes
// of a wave; as a per-lane POINTER the matrix would be read with 16 vector loads at the start of every pixel.  Instead the wave works
// through the eyes its lanes ask for one at a time (nearly always one): the choice is then uniform and the matrix comes through
// scalar loads.
// (`direction`: v is a ray direction, w = 0 -- every call but the one that makes the origin where the prologue kernel does not)
PTL_FN vec4 camera_times(const mat4& camera_matrix, int which_eye, vec4 v, bool direction = true) {
#if PTL_DEVICE_BUILD
    (void)camera_matrix;
    vec4 product = vec4(0.0f);
    for (bool done = false; !done;) {
        const int eye = __builtin_amdgcn_readfirstlane(which_eye);
        if (which_eye == eye) {
            const mat4& m = eye == 0 ? _camera : (eye == 1 ? _camera_left_eye : _camera_right_eye);
#ifdef PTL_AFFINE_RAYS
            // every caller hands over a DIRECTION (w = 0), and the renderer keeps this kernel only while the camera matrices are affine: the
            // translation column meets a zero, the w row gives that zero back (ptl_glsl.h `ptl_row_m`, PTL_W_ZERO)
            product = direction ? ptl_mul_runtime_direction(m, v) : ptl_mul_runtime(m, v);
#else
            product = ptl_mul_runtime(m, v);  // a run-time matrix in every build: the full chain, no zero tests (ptl_glsl.h `ptl_mterm`)
#endif
            done = true;
        }
    }
    (void)direction;
    return product;
#else
    (void)which_eye;
    (void)direction;
#ifdef PTL_AFFINE_RAYS
    return direction ? ptl_mul_runtime_direction(camera_matrix, v) : ptl_mul_runtime(camera_matrix, v);
#else
    return ptl_mul_runtime(camera_matrix, v);
#endif
#endif
}

#ifdef PTL_DERIVED_BUILTINS
// camera_matrix * (0, 0, 0, 1), which the prologue kernel has left behind the uniforms; read like camera_times reads the matrix.
PTL_FN vec4 camera_origin(int which_eye) {
#if PTL_DEVICE_BUILD
    vec4 o = vec4(0.0f);
    for (bool done = false; !done;) {
        const int eye = __builtin_amdgcn_readfirstlane(which_eye);
        if (which_eye == eye) {
            o = eye == 0 ? PTL_U.ptl_dv_origin : (eye == 1 ? PTL_U.ptl_dv_origin_left : PTL_U.ptl_dv_origin_right);
            done = true;
        }
    }
    return o;
#else
    return which_eye == 0 ? PTL_U.ptl_dv_origin : (which_eye == 1 ? PTL_U.ptl_dv_origin_left : PTL_U.ptl_dv_origin_right);
#endif
}
#endif

// Primary ray for one image-plane position, then trace it.  (src/frag.glsl:408-464)
// `which_eye` (0 the camera, 1 / 2 the left / right eye matrix) only tells where the prologue kernel has put the ray origin
// camera_matrix * (0, 0, 0, 1) of this frame.
PTL_FN vec3 get_color2(vec2 image_position, const mat4& camera_matrix, bool in_subspace, float camera_scale, vec2 resolution, int which_eye) {
    const float Pi = 3.14159265359f;
    const float Pi05 = Pi * 0.5f;
#ifdef PTL_DERIVED_BUILTINS
    vec4 o = camera_origin(which_eye);
#else
    vec4 o = camera_times(camera_matrix, which_eye, vec4(0.0f, 0.0f, 0.0f, 1.0f), false);
#endif
    vec4 d;
    if (_use_panini_projection == 1) {
        d = normalize(camera_times(camera_matrix, which_eye, vec4(PaniniProjection(vec2(image_position.x, image_position.y), _view_angle, _panini_param), 0.0f)));
    } else if (_use_360_camera == 1) {  // equirectangular, 2:1, black bars outside
        float coef = min(resolution.x, resolution.y);
        float ax = ptl_div(resolution.x, coef);
        float ay = ptl_div(resolution.y, coef);
        float rx;
        float ry;
        if (ax >= 2.0f * ay) {
            ry = ay;
            rx = 2.0f * ay;
        } else {
            rx = ax;
            ry = ptl_div(ax, 2.0f);
        }
        if (abs(image_position.x) > rx || abs(image_position.y) > ry) return vec3(0.0f);
        float yaw = ptl_div(image_position.x, rx) * Pi;
        float pitch = ptl_div(image_position.y, ry) * Pi05;
        vec3 dir_local = vec3(sin(yaw) * cos(pitch), sin(pitch), cos(yaw) * cos(pitch));
        d = normalize(camera_times(camera_matrix, which_eye, vec4(dir_local, 0.0f)));
    } else if (_use_180_camera == 1) {  // VR180 front hemisphere
        if (abs(image_position.x) > 1.0f || abs(image_position.y) > 1.0f) return vec3(0.0f);
        float yaw = image_position.x * Pi05;
        float pitch = image_position.y * Pi05;
        vec3 dir_local = vec3(sin(yaw) * cos(pitch), sin(pitch), cos(yaw) * cos(pitch));
        d = normalize(camera_times(camera_matrix, which_eye, vec4(dir_local, 0.0f)));
    } else {  // pinhole
#ifdef PTL_DERIVED_BUILTINS
        float h = PTL_U.ptl_dv_tan_half_view;
#else
        float h = tan(ptl_div(_view_angle, 2.0f));
#endif
        d = normalize(camera_times(camera_matrix, which_eye, vec4(image_position.x * h, image_position.y * h, 1.0f, 0.0f)));
    }

#ifdef PTL_FIRST_TRIP
    RayTraceResult trace = ray_tracing(Ray{o, d, 1.0f, in_subspace}, camera_scale, which_eye == 0);
#else
    RayTraceResult trace = ray_tracing(Ray{o, d, 1.0f, in_subspace}, camera_scale);
#endif
    if (_draw_depth_map == 1) {
        if (trace.has_depth) return sample_depth_gradient(trace.depth);
        return vec3(0.0f);
    }
    return trace.color;
}

#ifdef PTL_ANAGLYPH
// Red/cyan anaglyph of the two eye images with ghosting compensation.  (src/frag.glsl:343-406; the reference strips these
// lines unless `disable_anaglyph` is switched off, src/main.rs:939 -- here they are compiled in by FLAG_ANAGLYPH.)
// mode 0: both eyes as luminance; mode 1: right eye keeps its green/blue hue.
PTL_FN vec3 anaglyphCombineLinear(vec3 leftLin, vec3 rightLin, int mode) {
    leftLin = clamp(leftLin, 0.0f, 1.0f);
    rightLin = clamp(rightLin, 0.0f, 1.0f);
    const vec3 LUMA = vec3(0.299f, 0.587f, 0.114f);
    float P = _anaglyph_p;
    float Q = _anaglyph_q;
    float l = dot(leftLin, LUMA);
    float r = dot(rightLin, LUMA);
    float denom = max(1e-6f, 1.0f - P * Q);
    float Rout = ptl_div(l - P * r, denom);
    float Cout = ptl_div(r - Q * l, denom);
    if (mode == 0) return clamp(vec3(Rout, Cout, Cout), 0.0f, 1.0f);
    float sumGB = rightLin.g + rightLin.b;
    float k = (sumGB > 1e-6f) ? ptl_div(2.0f * Cout, sumGB) : 0.0f;
    return clamp(vec3(Rout, rightLin.g * k, rightLin.b * k), 0.0f, 1.0f);
}
#endif

// Mono / side-by-side / anaglyph selection.  (src/frag.glsl:466-503)
PTL_FN vec3 get_color(vec2 image_position) {
#ifdef PTL_ANAGLYPH
    if (_draw_anaglyph == 1) {
        return anaglyphCombineLinear(get_color2(image_position, _camera_left_eye, _left_eye_in_subspace == 1, _left_eye_scale, _resolution, 1),
                                     get_color2(image_position, _camera_right_eye, _right_eye_in_subspace == 1, _right_eye_scale, _resolution, 2),
                                     _anaglyph_mode);
    }
#endif
    mat4 final_matrix = _camera;
    bool final_in_subspace = _camera_in_subspace == 1;
    float final_scale = _camera_scale;
    vec2 final_resolution = _resolution;
    int final_eye = 0;

    if (_draw_side_by_side == 1) {
        float coef = min(_resolution.x, _resolution.y);
        vec2 position = image_position / 2.0f * coef + _resolution / 2.0f;
        vec2 resolution = vec2(ptl_div(_resolution.x, 2.0f), _resolution.y);
        float coef2 = min(resolution.x, resolution.y);
        if (position.x < resolution.x) {
            image_position = (position - resolution / 2.0f) / coef2 * 2.0f;
            final_matrix = _camera_left_eye;
            final_in_subspace = _left_eye_in_subspace == 1;
            final_scale = _left_eye_scale;
            final_eye = 1;
        } else {
            image_position = (position - vec2(resolution.x, 0.0f) - resolution / 2.0f) / coef2 * 2.0f;
            final_matrix = _camera_right_eye;
            final_in_subspace = _right_eye_in_subspace == 1;
            final_scale = _right_eye_scale;
            final_eye = 2;
        }
        final_resolution = resolution;
    }
    return get_color2(image_position, final_matrix, final_in_subspace, final_scale, final_resolution, final_eye);
}

// R2 low-discrepancy sub-pixel offsets.  (src/frag.glsl:506-513)
PTL_FN vec2 quasi_random(int i) {
    const float a1 = 0.7548776662466927600500267982588025643670318456949186300834636687f;
    const float a2 = 0.5698402909980532659121818632752155853637566123932930564053138358f;
    return vec2(mod(0.5f + a1 * float(i), 1.0f), mod(0.5f + a2 * float(i), 1.0f));
}

// One pixel: `position` is the pixel centre in pixel units, y down, (0,0) = top-left corner
// (vertex stage src/gui/scene.rs:1674-1697 evaluated at the fragment centre; AA loop and
// gamma-2 encode src/frag.glsl:515-527,550-551).  Returns the RGBA the reference writes to
// FragColor, before the GL RGBA8 conversion.
PTL_FN vec4 shade_pixel(vec2 position) {
    PTL_PIXEL_RELAUNDER();  // the per-pixel part reads its builtins (camera matrix, projection) through scalar loads as well
    float coef = min(_resolution.x, _resolution.y);
#ifdef PTL_DERIVED_BUILTINS
    vec2 uv_screen = (position - PTL_U.ptl_dv_half_resolution) / coef * 2.0f;
    float pixel_size = PTL_U.ptl_dv_pixel_size;
#else
    vec2 uv_screen = (position - _resolution / 2.0f) / coef * 2.0f;
    float pixel_size = ptl_rcp(min(_resolution.x, _resolution.y));
#endif
    vec3 result = vec3(0.0f);
    const int aa_count = _aa_count, aa_end = aa_count + _aa_start;  // read once: later reads would go through whatever the bounce loop left
    for (int a = _aa_start; a < aa_end; a++) {
        PTL_PIXEL_RELAUNDER();
        vec2 offset = quasi_random(a);
        result += get_color(uv_screen + offset * pixel_size * 2.0f);
    }
    result = sqrt(result / float(aa_count));
    return vec4(result, 1.0f);
}

};  // struct ptl_tracer
#undef PTL_U
#define PTL_U ptl_u

PTL_FN vec4 shade_pixel(vec2 position) {
    ptl_tracer t{&ptl_u};
    return t.shade_pixel(position);
}
PTL_FN void teleport_external_ray_entry(float* out6) {
    ptl_tracer t{&ptl_u};
    t.teleport_external_ray_entry(out6);
}
PTL_FN void derive_uniforms(ptl_uniform_block* block) {
    ptl_tracer t{&ptl_u};
    t.derive(block);
}

// GL fixed-point conversion of one channel: clamp to [0,1], scale, round to nearest.
PTL_FN unsigned int unorm8(float v) {
    if (!(v > 0.0f)) return 0u;  // also NaN
    if (v >= 1.0f) return 255u;
    return (unsigned int)floor(fma(v, 255.0f, 0.5f));
}
PTL_FN unsigned int pack_rgba8(vec4 c) {
    return unorm8(c.x) | (unorm8(c.y) << 8) | (unorm8(c.z) << 16) | (unorm8(c.w) << 24);
}

}  // namespace glsl
