// flac.hpp — C++ host layer over the C ABI of libflacgpu, shaped like the reference Go package `flac`
// (flacmeta.go): the same type and field names, the same block numbering, Read() leaving the stream on
// byte 0 of audio frame 0, errors as returned strings where Go returns `error` (empty = nil).
//
// Every block kind of the reference is carried with its payload (STREAMINFO, SEEKTABLE, PADDING,
// VORBIS_COMMENT, APPLICATION, CUESHEET, PICTURE); all parsing is done by the C ABI.
//
// The frame engine itself (Decoder) has no counterpart in the reference, which stops at the first frame
// (flacmeta.go:845-849); it takes what Read left behind.
#pragma once

#include <cstdint>
#include <cstdio>
#include <istream>
#include <string>
#include <vector>

#include "../../include/flacgpu.h"

namespace flac {

// MetadataBlockType, flacmeta.go:25-37: FlacSignature takes value 0, so STREAMINFO = 1 ... PICTURE = 7.
enum MetadataBlockType : uint32_t {
    FlacSignature = 0,
    MetadataStreaminfo = 1,
    MetadataPadding = 2,
    MetadataApplication = 3,
    MetadataSeektable = 4,
    MetadataVorbisComment = 5,
    MetadataCuesheet = 6,
    MetadataPicture = 7,
    MetadataInvalid = 127,
};

// (MetadataBlockType).String, flacmeta.go:174-194
inline const char *String(MetadataBlockType t)
{
    switch (t) {
    case MetadataStreaminfo: return "STREAMINFO";
    case MetadataPadding: return "PADDING";
    case MetadataApplication: return "APPLICATION";
    case MetadataSeektable: return "SEEKTABLE";
    case MetadataVorbisComment: return "VORBIS_COMMENT";
    case MetadataCuesheet: return "CUESHEET";
    case MetadataPicture: return "PICTURE";
    default: return "INVALID";
    }
}

struct MetadataBlockHeader {   // flacmeta.go:235-240
    MetadataBlockType Type = MetadataInvalid;
    uint32_t Length = 0;
    bool Last = false;
    bool operator==(const MetadataBlockHeader &o) const { return Type == o.Type && Length == o.Length && Last == o.Last; }
};

struct SeekpointBlock {        // flacmeta.go:256-260
    uint64_t SampleNumber = 0;
    uint64_t Offset = 0;
    uint16_t FrameSamples = 0;
    bool operator==(const SeekpointBlock &o) const
    {
        return SampleNumber == o.SampleNumber && Offset == o.Offset && FrameSamples == o.FrameSamples;
    }
};

struct StreaminfoBlock {       // flacmeta.go:263-273
    uint16_t MinBlockSize = 0, MaxBlockSize = 0;
    uint32_t MinFrameSize = 0, MaxFrameSize = 0;
    uint32_t SampleRate = 0;
    uint8_t Channels = 0, BitsPerSample = 0;
    uint64_t TotalSamples = 0;
    std::string MD5Signature;  // 32 hex digits, as the reference stores it (flacmeta.go:707-715)
    bool operator==(const StreaminfoBlock &o) const
    {
        return MinBlockSize == o.MinBlockSize && MaxBlockSize == o.MaxBlockSize && MinFrameSize == o.MinFrameSize &&
               MaxFrameSize == o.MaxFrameSize && SampleRate == o.SampleRate && Channels == o.Channels &&
               BitsPerSample == o.BitsPerSample && TotalSamples == o.TotalSamples && MD5Signature == o.MD5Signature;
    }
};

struct VorbisCommentBlock {    // flacmeta.go:276-280
    std::string Vendor;
    uint32_t TotalComments = 0;
    std::vector<std::string> Comments;
    bool operator==(const VorbisCommentBlock &o) const
    {
        return Vendor == o.Vendor && TotalComments == o.TotalComments && Comments == o.Comments;
    }
};

struct ApplicationBlock {      // flacmeta.go:197-200
    uint32_t Id = 0;
    std::vector<uint8_t> Data;
};

struct TrackIndex {            // flacmeta.go:228-232
    uint64_t SampleOffset = 0;
    uint8_t IndexPoint = 0;
    bool operator==(const TrackIndex &o) const { return SampleOffset == o.SampleOffset && IndexPoint == o.IndexPoint; }
};

struct CuesheetTrack {         // flacmeta.go:214-225
    uint64_t Offset = 0;
    uint8_t Number = 0;
    std::string ISRC;          // 12 bytes, NULs kept
    uint8_t Type = 0;
    bool PreEmphasis = false;
    uint8_t IndexPoints = 0;
    std::vector<TrackIndex> Indexes;
    bool operator==(const CuesheetTrack &o) const
    {
        return Offset == o.Offset && Number == o.Number && ISRC == o.ISRC && Type == o.Type && PreEmphasis == o.PreEmphasis &&
               IndexPoints == o.IndexPoints && Indexes == o.Indexes;
    }
};

struct CuesheetBlock {         // flacmeta.go:203-211
    std::string MediaCatalogNumber;   // 128 bytes, NUL padding kept (flacmeta.go:403)
    uint64_t LeadinSamples = 0;
    bool IsCompactDisc = false;
    uint8_t TotalTracks = 0;
    std::vector<CuesheetTrack> Tracks;
};

struct PictureBlock {          // flacmeta.go:243-253
    std::string PictureType, MimeType, Description;
    uint32_t Width = 0, Height = 0, ColorDepth = 0, NumColors = 0, Length = 0;
    std::vector<uint8_t> PictureBlob;
};

// Complete blocks (header + data), flacmeta.go:288-334.
struct Application { MetadataBlockHeader Header; ApplicationBlock Data; bool IsPopulated = false; };
struct Cuesheet { MetadataBlockHeader Header; CuesheetBlock Data; bool IsPopulated = false; };
struct Picture { MetadataBlockHeader Header; PictureBlock Data; bool IsPopulated = false; };
struct Padding { MetadataBlockHeader Header; bool IsPopulated = false; };
struct Streaminfo { MetadataBlockHeader Header; StreaminfoBlock Data; bool IsPopulated = false; };
struct VorbisComment { MetadataBlockHeader Header; VorbisCommentBlock Data; bool IsPopulated = false; };
struct Seektable {
    MetadataBlockHeader Header;
    std::vector<SeekpointBlock> Data;
    bool IsPopulated = false;
    int TotalPoints() const { return (int)Data.size(); }   // flacmeta.go:607-609
};

inline std::string hex(const uint8_t *b, size_t n)
{
    static const char *d = "0123456789abcdef";
    std::string s;
    for (size_t i = 0; i < n; ++i) { s += d[b[i] >> 4]; s += d[b[i] & 15]; }
    return s;
}

// Metadata, flacmeta.go:337-346.
struct Metadata {
    flac::Streaminfo Streaminfo;
    flac::Application Application;
    flac::VorbisComment VorbisComment;
    std::vector<flac::Picture> Pictures;
    flac::Padding Padding;
    flac::Seektable Seektable;
    flac::Cuesheet Cuesheet;
    std::vector<MetadataBlockHeader> Blocks;   // every block in file order, payloads not kept
    uint8_t TotalBlocks = 0;
    uint64_t FirstFrameOffset = 0;             // where Read leaves the stream

    // (*Metadata).Read, flacmeta.go:749-850.  Returns "" on success, otherwise the error text.  On success
    // `f` is positioned on the first byte of the first audio frame.
    std::string Read(std::istream &f)
    {
        // the metadata area is read block by block, like the reference does: 4 bytes of signature,
        // then a 4-byte header and its payload per block
        std::vector<uint8_t> buf(4);
        if (!f.read(reinterpret_cast<char *>(buf.data()), 4)) return "error reading FLAC signature: unexpected EOF";
        if (std::string(buf.begin(), buf.end()) != "fLaC") return "\"" + std::string(buf.begin(), buf.end()) + "\" is not a valid FLAC signature";
        for (;;) {
            const size_t at = buf.size();
            buf.resize(at + 4);
            if (!f.read(reinterpret_cast<char *>(buf.data() + at), 4)) return "error reading metadata block header: unexpected EOF";
            flacgpu_block hb;
            if (int rc = flacgpu_parse_block_header(buf.data() + at, 4, &hb))
                return std::string("failed to marshalMetadataBlockHeader: ") + flacgpu_status_string(rc);
            buf.resize(at + 4 + hb.length);
            if (hb.length && !f.read(reinterpret_cast<char *>(buf.data() + at + 4), hb.length)) {
                char msg[160];
                std::snprintf(msg, sizeof msg, "failed to read %lld of %u bytes for %s metadata block: unexpected EOF",
                              (long long)f.gcount(), hb.length, String((MetadataBlockType)hb.type));
                return msg;
            }
            if (hb.last) break;
        }
        flacgpu_metadata md;
        if (int rc = flacgpu_parse_metadata(buf.data(), buf.size(), &md)) {
            if (rc == FLACGPU_ERR_DUPLICATE) return "two blocks of a kind that may appear once encountered";
            if (rc == FLACGPU_ERR_SEEKTABLE) return "SEEKTABLE block length is not a multiple of 18";
            if (rc == FLACGPU_ERR_CUESHEET) return "invalid CUESHEET block";
            if (rc == FLACGPU_ERR_APPLICATION) return "malformed ApplictionBlock; field length not a multiple of 8";
            return std::string("metadata: ") + flacgpu_status_string(rc);
        }
        FirstFrameOffset = md.first_frame_offset;
        TotalBlocks = (uint8_t)md.n_blocks;
        for (uint32_t i = 0; i < md.n_blocks; ++i) {
            const flacgpu_block &b = md.blocks[i];
            MetadataBlockHeader h;
            h.Type = (MetadataBlockType)b.type; h.Length = b.length; h.Last = b.last != 0;
            Blocks.push_back(h);
            const uint8_t *p = buf.data() + b.payload_offset;
            switch (h.Type) {
            case MetadataStreaminfo: {
                Streaminfo.Header = h; Streaminfo.IsPopulated = true;
                const flacgpu_streaminfo &s = md.streaminfo;
                Streaminfo.Data.MinBlockSize = s.min_blocksize; Streaminfo.Data.MaxBlockSize = s.max_blocksize;
                Streaminfo.Data.MinFrameSize = s.min_framesize; Streaminfo.Data.MaxFrameSize = s.max_framesize;
                Streaminfo.Data.SampleRate = s.sample_rate; Streaminfo.Data.Channels = s.channels;
                Streaminfo.Data.BitsPerSample = s.bits_per_sample; Streaminfo.Data.TotalSamples = s.total_samples;
                Streaminfo.Data.MD5Signature = hex(s.md5, 16);
                break;
            }
            case MetadataPadding: Padding.Header = h; Padding.IsPopulated = true; break;
            case MetadataSeektable: {
                Seektable.Header = h; Seektable.IsPopulated = true;
                std::vector<flacgpu_seekpoint> pts(b.length / 18 + 1);
                size_t n = 0;
                if (int rc = flacgpu_parse_seektable(p, b.length, pts.data(), pts.size(), &n))
                    return std::string("SEEKTABLE: ") + flacgpu_status_string(rc);
                for (size_t k = 0; k < n; ++k) {
                    SeekpointBlock sp;
                    sp.SampleNumber = pts[k].sample_number; sp.Offset = pts[k].offset; sp.FrameSamples = pts[k].frame_samples;
                    Seektable.Data.push_back(sp);
                }
                break;
            }
            case MetadataVorbisComment: {
                // MarshalVorbisCommentBlock, flacmeta.go:720-746
                VorbisComment.Header = h; VorbisComment.IsPopulated = true;
                flacgpu_vorbis_comment v;
                size_t n = 0;
                if (int rc = flacgpu_parse_vorbis_comment(p, b.length, &v, nullptr, nullptr, 0, &n))
                    return std::string("VORBIS_COMMENT: ") + flacgpu_status_string(rc);
                std::vector<uint32_t> off(n + 1), len(n + 1);
                if (int rc = flacgpu_parse_vorbis_comment(p, b.length, &v, off.data(), len.data(), n, &n))
                    return std::string("VORBIS_COMMENT: ") + flacgpu_status_string(rc);
                VorbisComment.Data.Vendor.assign(reinterpret_cast<const char *>(p + v.vendor_offset), v.vendor_length);
                VorbisComment.Data.TotalComments = v.total_comments;
                for (size_t k = 0; k < n; ++k)
                    VorbisComment.Data.Comments.emplace_back(reinterpret_cast<const char *>(p + off[k]), len[k]);
                break;
            }
            case MetadataApplication: {
                // MarshalApplicationBlock, flacmeta.go:349-366
                Application.Header = h; Application.IsPopulated = true;
                flacgpu_application a;
                if (int rc = flacgpu_parse_application(p, b.length, &a)) return std::string("APPLICATION: ") + flacgpu_status_string(rc);
                Application.Data.Id = a.id;
                Application.Data.Data.assign(p + a.data_offset, p + a.data_offset + a.data_length);
                break;
            }
            case MetadataPicture: {
                // MarshalPictureBlock, flacmeta.go:554-604; pictures may repeat (flacmeta.go:799-801)
                flacgpu_picture pic;
                if (int rc = flacgpu_parse_picture(p, b.length, &pic)) return std::string("PICTURE: ") + flacgpu_status_string(rc);
                flac::Picture pc;
                pc.Header = h; pc.IsPopulated = true;
                pc.Data.PictureType = flacgpu_picture_type_string(pic.picture_type);
                pc.Data.MimeType.assign(reinterpret_cast<const char *>(p + pic.mime_offset), pic.mime_length);
                pc.Data.Description.assign(reinterpret_cast<const char *>(p + pic.description_offset), pic.description_length);
                pc.Data.Width = pic.width; pc.Data.Height = pic.height; pc.Data.ColorDepth = pic.color_depth;
                pc.Data.NumColors = pic.num_colors; pc.Data.Length = pic.length;
                pc.Data.PictureBlob.assign(p + pic.blob_offset, p + pic.blob_offset + pic.blob_length);
                Pictures.push_back(pc);
                break;
            }
            case MetadataCuesheet: {
                // MarshalCuesheetBlock / Track / TrackIndex, flacmeta.go:369-516
                Cuesheet.Header = h; Cuesheet.IsPopulated = true;
                flacgpu_cuesheet c;
                size_t ni = 0;
                if (int rc = flacgpu_parse_cuesheet(p, b.length, &c, nullptr, 0, nullptr, 0, &ni))
                    return std::string("CUESHEET: ") + flacgpu_status_string(rc);
                std::vector<flacgpu_cuesheet_track> tr(c.total_tracks);
                std::vector<flacgpu_cuesheet_index> ix(ni + 1);
                if (int rc = flacgpu_parse_cuesheet(p, b.length, &c, tr.data(), tr.size(), ix.data(), ix.size(), &ni))
                    return std::string("CUESHEET: ") + flacgpu_status_string(rc);
                CuesheetBlock &d = Cuesheet.Data;
                d.MediaCatalogNumber.assign(c.media_catalog_number, 128);
                d.LeadinSamples = c.leadin_samples; d.IsCompactDisc = c.is_compact_disc != 0; d.TotalTracks = c.total_tracks;
                for (const flacgpu_cuesheet_track &t : tr) {
                    CuesheetTrack ct;
                    ct.Offset = t.offset; ct.Number = t.number; ct.ISRC.assign(t.isrc, 12); ct.Type = t.type;
                    ct.PreEmphasis = t.pre_emphasis != 0; ct.IndexPoints = t.index_points;
                    for (uint32_t j = 0; j < t.index_points; ++j) {
                        TrackIndex ti;
                        ti.SampleOffset = ix[t.first_index + j].sample_offset; ti.IndexPoint = ix[t.first_index + j].index_point;
                        ct.Indexes.push_back(ti);
                    }
                    d.Tracks.push_back(ct);
                }
                break;
            }
            default: break;
            }
        }
        return "";
    }
};

// Result of decoding one stream (flacgpu_stream_result).
struct StreamResult {
    uint64_t HeaderCandidates = 0, Frames = 0, FramesOK = 0, Samples = 0, PCMBytes = 0;
    uint64_t FirstBadFrame = ~0ull;   // index of the first frame that did not decode, ~0 when every frame did
    int32_t FirstError = 0;           // its FLACGPU_FRAME_* status
    uint32_t MD5Status = FLACGPU_MD5_NOT_CHECKED;   // PCM against StreaminfoBlock.MD5Signature (flacmeta.go:664)
};

// Name of a per-frame status, for error texts in the reference's style (it names what failed, flacmeta.go:756-788).
inline const char *FrameStatusString(int32_t status)
{
    static const char *const names[] = {"ok", "CRC-16 mismatch", "invalid subframe", "frame truncated", "frame length mismatch",
                                        "samples outside the stream or the output", "unsupported", "inside another frame"};
    return status >= 0 && status < 8 ? names[status] : "unknown";
}

// One GPU's frame engine (flacgpu_create / flacgpu_destroy).
class Decoder {
public:
    Decoder() = default;
    Decoder(const Decoder &) = delete;
    Decoder &operator=(const Decoder &) = delete;
    ~Decoder() { Close(); }
    // NewDecoder: returns "" or the error text
    std::string Open(int device, bool verifyMD5 = true)
    {
        if (flacgpu_api_version() != FLACGPU_API_VERSION) return "libflacgpu was built from another version of flacgpu.h";
        flacgpu_options o;
        o.struct_size = sizeof o; o.device = device; o.flags = verifyMD5 ? FLACGPU_FLAG_VERIFY_MD5 : 0;
        o.output_format = FLACGPU_OUT_PACKED;
        if (int rc = flacgpu_create(&o, &h_)) return std::string("flacgpu_create: ") + flacgpu_status_string(rc);
        return "";
    }
    void Close() { if (h_) flacgpu_destroy(h_); h_ = nullptr; }

    // Decodes the frames that follow the metadata: `f` as Read left it.  pcm receives the interleaved
    // little-endian samples the STREAMINFO MD5 is defined over.
    std::string DecodeFrames(const Metadata &m, std::istream &f, std::vector<uint8_t> &pcm, StreamResult &res)
    {
        if (!h_) return "decoder is not open";
        if (!m.Streaminfo.IsPopulated) return "DecodeFrames needs the STREAMINFO that Read produced";
        std::vector<uint8_t> audio((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        const StreaminfoBlock &s = m.Streaminfo.Data;
        flacgpu_streaminfo si;
        si.min_blocksize = s.MinBlockSize; si.max_blocksize = s.MaxBlockSize;
        si.min_framesize = s.MinFrameSize; si.max_framesize = s.MaxFrameSize;
        si.sample_rate = s.SampleRate; si.channels = s.Channels; si.bits_per_sample = s.BitsPerSample;
        si.total_samples = s.TotalSamples;
        for (int i = 0; i < 16; ++i) si.md5[i] = (uint8_t)std::stoi(s.MD5Signature.substr(2 * i, 2), nullptr, 16);
        // The output is sized from STREAMINFO.  "0 == Implied Unknown" (flacmeta.go:662) leaves nothing to size it
        // from, and a hostile 36-bit TotalSamples would ask for terabytes: a frame is at least 8 + Channels bytes
        // (header, one constant subframe per channel, CRC-16) and holds at most MaxBlockSize samples.
        if (s.TotalSamples == 0)
            return "DecodeFrames: TotalSamples is unknown (0); scan the frames and decode with an explicit capacity";
        const uint64_t max_frames = audio.size() / (8u + s.Channels);
        if (s.TotalSamples > max_frames * (uint64_t)s.MaxBlockSize)
            return "DecodeFrames: STREAMINFO TotalSamples exceeds what the audio bytes can hold";
        pcm.assign((size_t)(s.TotalSamples * s.Channels * ((s.BitsPerSample + 7) / 8)), 0);
        const uint8_t *in = audio.data();
        const uint64_t len = audio.size(), cap = pcm.size();
        uint8_t *out = pcm.data();
        flacgpu_stream_result r;
        if (int rc = flacgpu_decode_batch(h_, &in, &len, &si, 1, &out, &cap, &r))
            return std::string("decode: ") + flacgpu_status_string(rc) + ": " + flacgpu_last_error(h_);
        res.HeaderCandidates = r.header_candidates; res.Frames = r.frames; res.FramesOK = r.frames_ok;
        res.Samples = r.samples; res.PCMBytes = r.pcm_bytes;
        res.FirstBadFrame = r.first_bad_frame; res.FirstError = r.first_error; res.MD5Status = r.md5_status;
        if (r.frames && !r.frames_ok) return "decode: no frame of the stream decoded";
        if (r.frames_ok < r.frames)    // PCM of the other frames is in place; the caller decides whether that is an error
            return std::string("decode: frame ") + std::to_string(r.first_bad_frame) + ": " + FrameStatusString(r.first_error);
        if (r.md5_status == FLACGPU_MD5_MISMATCH) return "decode: PCM does not match the STREAMINFO MD5 signature";
        return "";
    }

    // Options of EncodeBatch (flacgpu_encode_options).
    struct EncodeOptions {
        uint32_t BlockSize = 4096, MaxLPCOrder = 12, QLPPrecision = 0, MinPartitionOrder = 0, MaxPartitionOrder = 6;
        int32_t StereoMode = FLACGPU_STEREO_EXACT;   // FLACGPU_STEREO_ESTIMATE: about half the analysis work
        uint32_t ForceLPCOrder = 0;
        bool MD5OnDevice = true;
    };

    // Encodes interleaved little-endian PCM (ceil(bps/8) bytes per sample, what DecodeFrames returns) into
    // complete .flac streams that Metadata::Read accepts.  infos supplies SampleRate, Channels, BitsPerSample
    // and TotalSamples of every stream.
    std::string EncodeBatch(const std::vector<std::vector<uint8_t>> &pcm, const std::vector<StreaminfoBlock> &infos,
                            const EncodeOptions &o, std::vector<std::vector<uint8_t>> &files)
    {
        if (!h_) return "decoder is not open";
        const size_t n = pcm.size();
        if (n == 0 || n != infos.size()) return "EncodeBatch: streams and STREAMINFO blocks differ in number";
        flacgpu_encode_options opts = {};
        opts.struct_size = sizeof opts; opts.blocksize = o.BlockSize; opts.max_lpc_order = o.MaxLPCOrder;
        opts.qlp_precision = o.QLPPrecision; opts.min_partition_order = o.MinPartitionOrder;
        opts.max_partition_order = o.MaxPartitionOrder; opts.stereo_mode = o.StereoMode;
        opts.force_lpc_order = o.ForceLPCOrder; opts.flags = o.MD5OnDevice ? FLACGPU_ENC_MD5 : 0;
        std::vector<flacgpu_pcm_stream> streams(n);
        std::vector<const uint8_t *> in(n);
        std::vector<uint8_t *> out(n);
        std::vector<uint64_t> caps(n);
        std::vector<flacgpu_encode_result> res(n);
        files.assign(n, {});
        for (size_t i = 0; i < n; ++i) {
            const StreaminfoBlock &s = infos[i];
            flacgpu_pcm_stream &ps = streams[i];
            ps = flacgpu_pcm_stream{};
            ps.total_samples = s.TotalSamples; ps.sample_rate = s.SampleRate;
            ps.channels = s.Channels; ps.bits_per_sample = s.BitsPerSample;
            if (pcm[i].size() != s.TotalSamples * s.Channels * ((s.BitsPerSample + 7u) / 8u)) return "EncodeBatch: PCM size does not match TotalSamples";
            caps[i] = flacgpu_encode_bound(s.TotalSamples, s.Channels, s.BitsPerSample, o.BlockSize);
            files[i].resize((size_t)caps[i]);
            in[i] = pcm[i].data(); out[i] = files[i].data();
        }
        if (int rc = flacgpu_encode_batch(h_, in.data(), streams.data(), (uint32_t)n, &opts, out.data(), caps.data(), res.data()))
            return std::string("encode: ") + flacgpu_status_string(rc) + ": " + flacgpu_last_error(h_);
        for (size_t i = 0; i < n; ++i) {
            if (res[i].status) return std::string("encode: stream ") + std::to_string(i) + ": " + flacgpu_status_string(res[i].status);
            files[i].resize((size_t)res[i].out_length);
        }
        return "";
    }

private:
    flacgpu_handle *h_ = nullptr;
};

}  // namespace flac
