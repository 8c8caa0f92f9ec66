// ref_shim_post.cpp -- TEST INFRASTRUCTURE ONLY.  Appended after the extracted reference text:
// C entry points that load an ntscsim_params snapshot into the reference's globals
// (ffmpeg_ntsc.cpp:205-214, :756-809) and call its composite_layer() (:1570), and one that runs the
// field loop's bob block (:2233-2257, wrapped as ref_field_deinterlace by build_ref.sh) on a frame.
#include "ntscsim.h"
extern "C" void ntsc_ref_set_params(const ntscsim_params *p)
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
    vhs_head_switching = p->vhs_head_switching != 0;
    vhs_head_switching_point = p->vhs_head_switching_point;
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
}
extern "C" void ntsc_ref_srand(unsigned seed) { srand(seed); }
extern "C" unsigned ntsc_ref_rand(void) { return (unsigned)rand(); }
extern "C" void ntsc_ref_composite_layer(uint8_t *dst, int dst_linesize,
                                         const uint8_t *src, int src_linesize,
                                         int src_interlaced, int src_tff,
                                         int width, int height,
                                         unsigned field, unsigned long long fieldno)
{
    AVFrame d, s;
    InputFile dummy;
    memset(&d, 0, sizeof(d));
    memset(&s, 0, sizeof(s));
    d.data[0] = dst; d.linesize[0] = dst_linesize; d.width = width; d.height = height;
    s.data[0] = const_cast<uint8_t *>(src); s.linesize[0] = src_linesize;
    s.width = width; s.height = height;
    s.interlaced_frame = src_interlaced; s.top_field_first = src_tff;
    composite_layer(&d, &s, dummy, field, fieldno);
}

// the "field deinterlace" block of main()'s field loop on one BGRA frame; `current` is the loop's field
// counter (field = (current & 1) ^ 1, :2234)
extern "C" void ntsc_ref_bob(uint8_t *frame, int linesize, int width, int height, unsigned long long current)
{
    AVFrame f;
    memset(&f, 0, sizeof(f));
    f.data[0] = frame; f.linesize[0] = linesize; f.width = width; f.height = height;
    output_avstream_video_frame.assign(1, &f);
    output_avstream_video_frame_index = 0;
    ref_field_deinterlace(current);
    output_avstream_video_frame.clear();
}
