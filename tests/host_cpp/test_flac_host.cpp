// Mirrors the reference's own tests (flacmeta_test.go: TestParseMetadata44k16Mono :10-93,
// TestParseMetadata44k16Stereo :95-278) against the C++ host layer, and, with --gpu, decodes both fixtures
// and compares the MD5 of the PCM with the STREAMINFO signature the reference's tests assert.
// Usage: test_flac_host <testdata dir> [--gpu]
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

#include "../../flac_amd/host/flac.hpp"

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)

// MD5 (RFC 1321), for the signature check only.
static std::string md5hex(const std::vector<uint8_t> &msg)
{
    static const uint32_t K[64] = {
        0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af,
        0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa,
        0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8,
        0x676f02d9, 0x8d2a4c8a, 0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
        0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97,
        0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1,
        0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
    static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20,
                              4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
    uint32_t a0 = 0x67452301, b0 = 0xefcdab89, c0 = 0x98badcfe, d0 = 0x10325476;
    std::vector<uint8_t> m(msg);
    const uint64_t bits = (uint64_t)msg.size() * 8;
    m.push_back(0x80);
    while (m.size() % 64 != 56) m.push_back(0);
    for (int i = 0; i < 8; ++i) m.push_back((uint8_t)(bits >> (8 * i)));
    for (size_t off = 0; off < m.size(); off += 64) {
        uint32_t w[16];
        for (int i = 0; i < 16; ++i) std::memcpy(&w[i], &m[off + 4 * i], 4);
        uint32_t A = a0, B = b0, C = c0, D = d0;
        for (int i = 0; i < 64; ++i) {
            uint32_t F; int g;
            if (i < 16) { F = (B & C) | (~B & D); g = i; }
            else if (i < 32) { F = (D & B) | (~D & C); g = (5 * i + 1) % 16; }
            else if (i < 48) { F = B ^ C ^ D; g = (3 * i + 5) % 16; }
            else { F = C ^ (B | ~D); g = (7 * i) % 16; }
            F += A + K[i] + w[g];
            A = D; D = C; C = B;
            B += (F << S[i]) | (F >> (32 - S[i]));
        }
        a0 += A; b0 += B; c0 += C; d0 += D;
    }
    uint8_t out[16];
    std::memcpy(out, &a0, 4); std::memcpy(out + 4, &b0, 4); std::memcpy(out + 8, &c0, 4); std::memcpy(out + 12, &d0, 4);
    return flac::hex(out, 16);
}

static flac::MetadataBlockHeader hdr(flac::MetadataBlockType t, uint32_t len, bool last)
{
    flac::MetadataBlockHeader h; h.Type = t; h.Length = len; h.Last = last; return h;
}

static void mono(const std::string &dir, bool gpu)
{
    std::ifstream f(dir + "/44100-16-mono.flac", std::ios::binary);
    CHECK(f.good());
    flac::Metadata got;
    const std::string err = got.Read(f);
    CHECK(err.empty());
    flac::StreaminfoBlock si;
    si.MinBlockSize = 4096; si.MaxBlockSize = 4096; si.MinFrameSize = 11; si.MaxFrameSize = 14; si.SampleRate = 44100;
    si.Channels = 1; si.BitsPerSample = 16; si.TotalSamples = 1014300; si.MD5Signature = "e5ccc967ced6c111530e5c79e33c969e";
    CHECK(got.Streaminfo.IsPopulated && got.Streaminfo.Header == hdr(flac::MetadataStreaminfo, 34, false) && got.Streaminfo.Data == si);
    flac::VorbisCommentBlock vc;
    vc.Vendor = "reference libFLAC 1.2.1 20070917"; vc.TotalComments = 1; vc.Comments = {"ARTIST=GoGoGo"};
    CHECK(got.VorbisComment.IsPopulated && got.VorbisComment.Header == hdr(flac::MetadataVorbisComment, 57, false) && got.VorbisComment.Data == vc);
    CHECK(got.Padding.IsPopulated && got.Padding.Header == hdr(flac::MetadataPadding, 8175, true));
    const std::vector<flac::SeekpointBlock> st = {{0, 0, 4096}, {438272, 1177, 4096}, {880640, 2452, 4096}};
    CHECK(got.Seektable.IsPopulated && got.Seektable.Header == hdr(flac::MetadataSeektable, 54, false) && got.Seektable.Data == st);
    CHECK(got.Seektable.TotalPoints() == 3);
    CHECK((uint64_t)f.tellg() == got.FirstFrameOffset && got.FirstFrameOffset == 8340);
    if (gpu) {
        flac::Decoder dec;
        CHECK(dec.Open(0).empty());
        std::vector<uint8_t> pcm;
        flac::StreamResult r;
        const std::string e = dec.DecodeFrames(got, f, pcm, r);
        CHECK(e.empty());
        CHECK(r.Frames == 248 && r.FramesOK == 248 && r.Samples == 1014300);
        CHECK(r.FirstBadFrame == ~0ull && r.FirstError == 0 && r.MD5Status == FLACGPU_MD5_MATCH);
        CHECK(md5hex(pcm) == si.MD5Signature);
    }
}

static void stereo(const std::string &dir, bool gpu)
{
    std::ifstream f(dir + "/silence-44-s.flac", std::ios::binary);
    CHECK(f.good());
    flac::Metadata got;
    CHECK(got.Read(f).empty());
    flac::StreaminfoBlock si;
    si.MinBlockSize = 4608; si.MaxBlockSize = 4608; si.MinFrameSize = 633; si.MaxFrameSize = 1323; si.SampleRate = 44100;
    si.Channels = 2; si.BitsPerSample = 16; si.TotalSamples = 162496; si.MD5Signature = "6291dbd8dcb7dc480132e4c4ba154a17";
    CHECK(got.Streaminfo.IsPopulated && got.Streaminfo.Header == hdr(flac::MetadataStreaminfo, 34, false) && got.Streaminfo.Data == si);
    flac::VorbisCommentBlock vc;
    vc.Vendor = "reference libFLAC 1.1.0 20030126"; vc.TotalComments = 7;
    vc.Comments = {"album=Quod Libet Test Data", "artist=piman", "artist=jzig", "genre=Silence", "tracknumber=02/10", "date=2004", "title=Silence"};
    CHECK(got.VorbisComment.IsPopulated && got.VorbisComment.Header == hdr(flac::MetadataVorbisComment, 169, false) && got.VorbisComment.Data == vc);
    const std::vector<flac::SeekpointBlock> st = {{0, 0, 4608}, {41472, 11852, 4608}, {50688, 14484, 4608}, {87552, 25022, 4608},
                                                  {105984, 30284, 4608}, {18446744073709551615ull, 0, 0}};
    CHECK(got.Seektable.IsPopulated && got.Seektable.Header == hdr(flac::MetadataSeektable, 108, false) && got.Seektable.Data == st);
    CHECK(got.Padding.IsPopulated && got.Padding.Header == hdr(flac::MetadataPadding, 3060, true));
    // CUESHEET, flacmeta_test.go:154-217
    {
        const flac::Cuesheet &c = got.Cuesheet;
        CHECK(c.IsPopulated && c.Header == hdr(flac::MetadataCuesheet, 588, false));
        std::string catalog = "1234567890123";
        catalog.resize(128, '\0');
        const std::string none(12, '\0');
        CHECK(c.Data.MediaCatalogNumber == catalog && c.Data.LeadinSamples == 0x15888 && c.Data.IsCompactDisc && c.Data.TotalTracks == 4);
        std::vector<flac::CuesheetTrack> want(4);
        want[0].Offset = 0; want[0].Number = 1; want[0].ISRC = "123456789012"; want[0].IndexPoints = 1; want[0].Indexes = {{0, 1}};
        want[1].Offset = 44100; want[1].Number = 2; want[1].ISRC = none; want[1].Type = 1; want[1].PreEmphasis = true;
        want[1].IndexPoints = 2; want[1].Indexes = {{0, 1}, {588, 2}};
        want[2].Offset = 88200; want[2].Number = 3; want[2].ISRC = none; want[2].IndexPoints = 1; want[2].Indexes = {{0, 1}};
        want[3].Offset = 162496; want[3].Number = 170; want[3].ISRC = none;
        CHECK(c.Data.Tracks == want);
    }
    // PICTURE, flacmeta_test.go:239-264
    {
        CHECK(got.Pictures.size() == 1);
        const flac::Picture &p = got.Pictures[0];
        CHECK(p.IsPopulated && p.Header == hdr(flac::MetadataPicture, 199, false));
        CHECK(p.Data.PictureType == "Cover (front)" && p.Data.MimeType == "image/png" && p.Data.Description == "A pixel.");
        CHECK(p.Data.Width == 1 && p.Data.Height == 1 && p.Data.ColorDepth == 24 && p.Data.NumColors == 0 && p.Data.Length == 150);
        CHECK(p.Data.PictureBlob.size() == 150 && p.Data.PictureBlob[0] == 0x89 && p.Data.PictureBlob[1] == 'P' &&
              p.Data.PictureBlob[148] == 0x60 && p.Data.PictureBlob[149] == 0x82);
        CHECK(flac::hex(p.Data.PictureBlob.data(), 8) == "89504e470d0a1a0a");
    }
    CHECK(!got.Application.IsPopulated);
    CHECK((uint64_t)f.tellg() == got.FirstFrameOffset && got.FirstFrameOffset == 4186);
    if (gpu) {
        flac::Decoder dec;
        CHECK(dec.Open(0).empty());
        std::vector<uint8_t> pcm;
        flac::StreamResult r;
        CHECK(dec.DecodeFrames(got, f, pcm, r).empty());
        CHECK(r.Frames == 36 && r.FramesOK == 36 && r.Samples == 162496);
        CHECK(r.FirstBadFrame == ~0ull && r.MD5Status == FLACGPU_MD5_MATCH);
        CHECK(md5hex(pcm) == si.MD5Signature);
        {
            // one flipped bit inside frame 3's residual: the error names the frame, the other frames decode
            std::ifstream h(dir + "/silence-44-s.flac", std::ios::binary);
            std::string bytes((std::istreambuf_iterator<char>(h)), std::istreambuf_iterator<char>());
            bytes[4186 + 1316 + 1319 + 1321 + 700] ^= 0x10;
            std::istringstream hs(bytes);
            flac::Metadata m2;
            CHECK(m2.Read(hs).empty());
            std::vector<uint8_t> pcm3;
            flac::StreamResult r3;
            const std::string e3 = dec.DecodeFrames(m2, hs, pcm3, r3);
            CHECK(r3.FirstBadFrame == 3 && r3.FirstError == FLACGPU_FRAME_BAD_CRC16 && r3.FramesOK == 35);
            CHECK(e3 == "decode: frame 3: CRC-16 mismatch");
        }
        // write side: the PCM just decoded goes through EncodeBatch, the result through Read and DecodeFrames again
        std::vector<std::vector<uint8_t>> files;
        flac::Decoder::EncodeOptions eo;
        eo.BlockSize = 4608;
        CHECK(dec.EncodeBatch({pcm}, {si}, eo, files).empty());
        CHECK(files.size() == 1 && files[0].size() > 42);
        std::istringstream g(std::string(files[0].begin(), files[0].end()));
        flac::Metadata again;
        CHECK(again.Read(g).empty());
        CHECK(again.FirstFrameOffset == 42 && again.Streaminfo.IsPopulated);
        const flac::StreaminfoBlock &t = again.Streaminfo.Data;
        CHECK(t.MinBlockSize == 4608 && t.MaxBlockSize == 4608 && t.SampleRate == si.SampleRate && t.Channels == si.Channels);
        CHECK(t.BitsPerSample == si.BitsPerSample && t.TotalSamples == si.TotalSamples && t.MD5Signature == si.MD5Signature);
        std::vector<uint8_t> pcm2;
        flac::StreamResult r2;
        CHECK(dec.DecodeFrames(again, g, pcm2, r2).empty());
        CHECK(r2.Frames == 36 && r2.FramesOK == 36 && pcm2 == pcm);
    }
}

static void errors()
{
    // not a FLAC signature (flacmeta.go:759-761); two STREAMINFO blocks (flacmeta.go:783-786)
    std::istringstream a(std::string("OggS") + std::string(60, '\0'));
    flac::Metadata m;
    CHECK(m.Read(a) == "\"OggS\" is not a valid FLAC signature");
    std::string two = "fLaC";
    const std::string si = std::string("\x00\x00\x00\x22", 4) + std::string("\x10\x00\x10\x00\x00\x00\x0b\x00\x00\x0e\x0a\xc4\x40\xf0\x00\x0f\x7a\x1c", 18) + std::string(16, '\x11');
    two += si + std::string("\x80\x00\x00\x22", 4) + si.substr(4);
    std::istringstream b(two);
    flac::Metadata m2;
    CHECK(!m2.Read(b).empty());
    std::istringstream c(std::string("fLaC\x00\x00\x00\x22", 8) + "short");
    flac::Metadata m3;
    CHECK(m3.Read(c).find("failed to read") == 0);
}

int main(int argc, char **argv)
{
    if (argc < 2) { std::printf("usage: %s <testdata> [--gpu]\n", argv[0]); return 2; }
    const bool gpu = argc > 2 && std::string(argv[2]) == "--gpu";
    mono(argv[1], gpu);
    stereo(argv[1], gpu);
    errors();
    if (failures) std::printf("%d checks failed\n", failures);
    else std::printf("all checks passed%s\n", gpu ? " (with GPU decode)" : "");
    return failures ? 1 : 0;
}
