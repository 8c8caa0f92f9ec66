// ref_tocomp_post.cpp -- TEST INFRASTRUCTURE ONLY.  C entry points around the extracted
// ffmpeg_to_composite.cpp text: globals from an ntscsim_params snapshot (:267-333), then
// composite_video_process (:629), render_field (:1001), black_key_feedback (:974).
#include "ntscsim.h"
extern "C" void tocomp_ref_set_params(const ntscsim_params *p)
{
    output_ntsc = (p->tv_standard == NTSCSIM_TV_NTSC);
    output_pal = !output_ntsc;
    output_width = p->output_width;
    output_height = p->output_height;
    video_scanline_phase_shift = p->video_scanline_phase_shift;
    video_scanline_phase_shift_offset = p->video_scanline_phase_shift_offset;
    composite_preemphasis = p->composite_preemphasis;
    composite_preemphasis_cut = p->composite_preemphasis_cut;
    vhs_out_sharpen = p->vhs_out_sharpen;
    vhs_out_sharpen_chroma = p->vhs_out_sharpen_chroma;
    vhs_head_switching = p->vhs_head_switching != 0;
    vhs_head_switching_phase = p->vhs_head_switching_phase;
    vhs_head_switching_phase_noise = p->vhs_head_switching_phase_noise;
    composite_in_chroma_lowpass = p->composite_in_chroma_lowpass != 0;
    composite_out_chroma_lowpass = p->composite_out_chroma_lowpass != 0;
    composite_out_chroma_lowpass_lite = p->composite_out_chroma_lowpass_lite != 0;
    video_yc_recombine = p->video_yc_recombine;
    video_chroma_noise = p->video_chroma_noise;
    video_chroma_phase_noise = p->video_chroma_phase_noise;
    video_chroma_loss = p->video_chroma_loss;
    video_noise = p->video_noise;
    subcarrier_amplitude = p->subcarrier_amplitude;
    subcarrier_amplitude_back = p->subcarrier_amplitude_back;
    emulating_vhs = p->emulating_vhs != 0;
    nocolor_subcarrier = p->nocolor_subcarrier != 0;
    nocolor_subcarrier_after_yc_sep = p->nocolor_subcarrier_after_yc_sep != 0;
    vhs_chroma_vert_blend = p->vhs_chroma_vert_blend != 0;
    vhs_svideo_out = p->vhs_svideo_out != 0;
    output_vhs_tape_speed = p->output_vhs_tape_speed;
    black_key_level_feedback = p->black_key_level_feedback;
}
static void fill(AVFrame &f, uint8_t *const *data, const int *ls, int w, int h)
{
    memset(&f, 0, sizeof(f));
    for (int i = 0; i < 3; i++) { f.data[i] = data[i]; f.linesize[i] = ls[i]; }
    f.width = w; f.height = h; f.format = AV_PIX_FMT_YUV422P;
}
extern "C" void tocomp_ref_srand(unsigned s) { srand(s); }
extern "C" void tocomp_ref_process(uint8_t *const *data, const int *ls, int w, int h,
                                   unsigned field, unsigned long long fieldno)
{
    AVFrame d; fill(d, data, ls, w, h);
    composite_video_process(&d, field, fieldno);
}
extern "C" void tocomp_ref_render_field(uint8_t *const *ddata, const int *dls, int w, int h,
                                        uint8_t *const *sdata, const int *sls, int sh, int is420,
                                        int interlaced, int tff, int second_field, unsigned field)
{
    AVFrame d, s; fill(d, ddata, dls, w, h); fill(s, sdata, sls, w, sh);
    s.interlaced_frame = interlaced; s.top_field_first = tff;
    output_avstream_video_input_frame->format = is420 ? AV_PIX_FMT_YUV420P : AV_PIX_FMT_YUV422P;
    // (field_number - src_pts) >= ticks_per_frame/2 <=> second field (:1033-1036)
    render_field(&d, &s, field, 100, second_field ? 99 : 100);
}
extern "C" void tocomp_ref_black_key_feedback(uint8_t *const *ddata, const int *dls,
                                              uint8_t *const *fdata, const int *fls, int w, int h,
                                              unsigned field)
{
    AVFrame d, f; fill(d, ddata, dls, w, h); fill(f, fdata, fls, w, h);
    black_key_feedback(&d, &f, field, 0);
}
extern "C" void tocomp_ref_output_frame(uint8_t *const *bdata, const int *bls,
                                        uint8_t *const *fdata, const int *fls, int w, int h,
                                        unsigned field, int mode)
{
    AVFrame b, f; fill(b, bdata, bls, w, h); fill(f, fdata, fls, w, h);
    use_422_colorspace = (mode == 0);
    output_video_as_interlaced = (mode == 2);
    output_avstream_video_bob_frame = &b;
    ref_output_bob(&f, field);
}
