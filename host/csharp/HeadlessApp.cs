// Headless counterpart of RayTracing.App (App.cs:7-68): no window (MI355X has no display engine); renders N frames and
// writes the last one as a binary PPM. NOT compiled here (no dotnet in the image).
using System;
using System.IO;
using RayTracing.Graphics;

namespace RayTracing;

public class HeadlessApp : IDisposable
{
    public HipRenderer Renderer { get; private set; } = null!;
    public int Frames = 1;
    public string Output = "frame.ppm";
    public PtSceneKind? Scene;      // null = the reference's own one-sphere frame
    public uint Detail, Spp = 64;

    public void Run()
    {
        InitRenderer();
        for (int i = 0; i < Frames; i++) Renderer.Render(0f); // the Window.Render event of App.cs:39-42
        byte[] px = Renderer.ReadFramebufferRgba8();
        using FileStream f = File.Create(Output);
        f.Write(System.Text.Encoding.ASCII.GetBytes($"P6\n{Renderer.Width} {Renderer.Height}\n255\n"));
        for (int i = 0; i < px.Length; i += 4) f.Write(px, i, 3);
        PtStats s = Renderer.LastStats;
        Console.WriteLine($"{s.rays} rays, {s.gpu_ms:F2} ms, {s.rays / s.gpu_ms / 1e3:F1} Mrays/s -> {Output}");
    }

    private void InitRenderer()
    {
        Renderer = new HipRenderer();
        Renderer.Init();
        if (Scene is PtSceneKind k) { Renderer.LoadSyntheticScene(k, Detail); Renderer.Params.spp = Spp; }
    }

    public void Dispose() { Renderer?.Dispose(); GC.SuppressFinalize(this); }

    public static void Main(string[] args)
    {
        using HeadlessApp app = new();
        for (int i = 0; i + 1 < args.Length; i += 2)
            switch (args[i])
            {
                case "--scene": app.Scene = Enum.Parse<PtSceneKind>(args[i + 1], true); break;
                case "--detail": app.Detail = uint.Parse(args[i + 1]); break;
                case "--spp": app.Spp = uint.Parse(args[i + 1]); break;
                case "--frames": app.Frames = int.Parse(args[i + 1]); break;
                case "--out": app.Output = args[i + 1]; break;
            }
        app.Run();
    }
}
