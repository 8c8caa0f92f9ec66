// ref_raw28_post.cpp -- TEST INFRASTRUCTURE ONLY.  Appended after the extracted ranges of
// ffmpeg_raw28ntsc.cpp: a C entry point that performs main()'s own set-up and field loop around the
// extracted functions.  main() itself cannot be extracted (it is interleaved with libav* calls), so
// the statements below RESTATE :866-897 (rate :866-875, compute_NTSC :876, preset_NTSC :877, delay line
// :886-887, detector filters :889-892, open_src :894) and :1006-1019 + :1037 (the loop body up to
// composite_layer, and current++); everything they call is the reference's text.
extern "C" {

struct raw28_ref_opts {
    double  sample_rate;            // 0 = ntsc28
    int32_t mark_sync, disable_sync, disable_wp_equ, show_subcarrier, disable_subcarrier, disable_equalization;
};

// Runs the tool's video path on the capture file `path`; writes up to max_fields BGRA frames of
// output_width x output_height (linesize = 4 * width, packed) into `frames`.  Returns the number of
// fields, stores the geometry and the final levels / read position.
int raw28_ref_run(const raw28_ref_opts *o, const char *path, uint8_t *frames, int max_fields,
                  int *width, int *height, int *scanline, double *blank_out, double *white_out,
                  unsigned long long *read_pos)
{
    // ---- state the tool initialises statically (so that the function can be called again)
    mark_sync = o->mark_sync != 0; disable_sync = o->disable_sync != 0; disable_wp_equ = o->disable_wp_equ != 0;
    show_subcarrier = o->show_subcarrier != 0; disable_subcarrier = o->disable_subcarrier != 0;
    disable_equalization = o->disable_equalization != 0;
    src_byte_counter = 0; close_src();
    std::vector<oneprocsamp>().swap(input_samples);      // a fresh process starts with an empty (then zero-filled) buffer
    memset(int_scanline, 0, sizeof(int_scanline)); memset(int_chroma, 0, sizeof(int_chroma));
    memset(int_luma, 0, sizeof(int_luma));
    hsync_dc_level = 128.0; blank_level = (uint8_t)0; white_level = (uint8_t)192;
    for (size_t i = 0; i < hsync_dc_detect_passes; i++) hsync_dc_detect[i] = LowpassFilter();
    src_composite.clear();
    src_composite.push_back(path);
    // ---- main() :866-897
    if (o->sample_rate > 0) sample_rate = o->sample_rate; else NTSC28MHz();
    compute_NTSC();
    preset_NTSC();
    hsync_dc_detect_delay.clear();
    hsync_dc_detect_delay.resize((size_t)((one_scanline_time * 0.075 * 0.75) * 0.5));
    hsync_dc_detect_delay_i = hsync_dc_detect_delay.begin();
    for (size_t i = 0; i < hsync_dc_detect_passes; i++) {
        hsync_dc_detect[i].setFilter(sample_rate, sample_rate / (one_scanline_time * 0.075 * 0.75));
        for (size_t j = 0; j < one_frame_time; j++) hsync_dc_detect[i].lowpass(128);
    }
    if (!open_src()) return -1;
    if (width) *width = output_width;
    if (height) *height = output_height;
    if (scanline) *scanline = (int)one_scanline_raw_length;
    AVFrame fr;
    memset(&fr, 0, sizeof(fr));
    fr.width = output_width; fr.height = output_height; fr.linesize[0] = output_width * 4;
    // ---- main() :1006-1019, :1037
    signed long long current = 0;
    int n = 0;
    while (n < max_fields) {
        lazy_flush_src();
        refill_src();
        if (count_src() < (one_scanline_raw_length * 256)) {
            close_src();
            if (!open_src()) break;
        }
        fr.data[0] = frames + (size_t)n * fr.linesize[0] * fr.height;
        memset(fr.data[0], 0, (size_t)fr.linesize[0] * fr.height);
        composite_layer(&fr, (current & 1) ^ 1, current);
        current++;
        n++;
    }
    if (blank_out) *blank_out = blank_level;
    if (white_out) *white_out = white_level;
    if (read_pos) *read_pos = total_count_src();
    close_src();
    return n;
}

// the front end alone: hsync_dc_proc() over a memory buffer
void raw28_ref_front(const raw28_ref_opts *o, const uint8_t *cap, size_t n, uint8_t *h, uint8_t *raw)
{
    mark_sync = o->mark_sync != 0;
    hsync_dc_level = 128.0;
    for (size_t i = 0; i < hsync_dc_detect_passes; i++) hsync_dc_detect[i] = LowpassFilter();
    if (o->sample_rate > 0) sample_rate = o->sample_rate; else NTSC28MHz();
    compute_NTSC();
    hsync_dc_detect_delay.clear();
    hsync_dc_detect_delay.resize((size_t)((one_scanline_time * 0.075 * 0.75) * 0.5));
    hsync_dc_detect_delay_i = hsync_dc_detect_delay.begin();
    for (size_t i = 0; i < hsync_dc_detect_passes; i++) {
        hsync_dc_detect[i].setFilter(sample_rate, sample_rate / (one_scanline_time * 0.075 * 0.75));
        for (size_t j = 0; j < one_frame_time; j++) hsync_dc_detect[i].lowpass(128);
    }
    for (size_t s = 0; s < n; s++) {
        oneprocsamp v;
        memset(&v, 0, sizeof(v));
        v.raw = cap[s];
        v = hsync_dc_proc(v);
        h[s] = v.hsync_dc_raw;
        raw[s] = v.raw;
    }
}

}
