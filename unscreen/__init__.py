"""Drop-in namespace: `from unscreen.<stage> import <Agent>` resolves to the MI355X-native implementations in
video_unscreen_amd (same class / function names as AnyiRao/video_unscreen's `unscreen` package)."""
